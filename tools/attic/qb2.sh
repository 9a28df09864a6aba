#!/bin/bash
# split walk vs fused walk on the GPU box: parity tests with the default (split) walk, then the bench line of both walks on both scenes
# usage (through gpurun): bash tools/qb2.sh <tag> [pytest -k expression]
tag=${1:-qb2}
timeout 600 python -m pytest tests/test_surfel_gpu.py -m gpu -q -x ${2:+-k "$2"} 2>&1 | tail -6
for fl in 0 4; do
for scene in surface stress; do
  GA_SURFEL_FLAGS=$fl timeout 300 python bench.py --no-dit --no-cpu-baseline --scene $scene > gpurun_out/${tag}_${scene}_$fl.json 2> gpurun_out/${tag}_${scene}_$fl.err || tail -5 gpurun_out/${tag}_${scene}_$fl.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_${scene}_$fl.json"))
print("flags $fl ${scene}", "ms/step", d["ms_per_step"], d["stage_ms"], "parity", d.get("parity", {}).get("max_mse"), {k: v for k, v in d["blend_valu"].items() if k != "note"})
PY
done
done
