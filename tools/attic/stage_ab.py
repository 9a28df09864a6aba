"""Stage times (HIP events) of the rasterizer forward with alternative builds of the library (tools/blend_variants.sh).
usage (GPU box): python tools/stage_ab.py name1 name2 ...   ('main' = the in-tree library)"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "gaussiananything_amd", "lib", "libga_mi355.so")


def main():
    backup = MAIN + ".ab_backup"
    shutil.copy(MAIN, backup)
    try:
        for name in sys.argv[1:] or ["main"]:
            shutil.copy(backup if name == "main" else os.path.join(ROOT, "tools", "_build", f"libga_{name}.so"), MAIN)
            for scene in ("surface", "stress"):
                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-dit", "--no-cpu-baseline", "--no-parity", "--scene", scene],
                                   capture_output=True, text=True, timeout=120)
                try:
                    import json
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    print(name, scene, "ms/step", d["ms_per_step"], d["stage_ms"], flush=True)
                except Exception:
                    print(name, scene, "FAILED", r.stderr[-500:], flush=True)
    finally:
        shutil.copy(backup, MAIN)
        os.remove(backup)


if __name__ == "__main__":
    main()
