"""CPU oracle for the render-and-denoise hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (gaussiananything_amd) never does.
"""
