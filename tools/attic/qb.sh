#!/bin/bash
# quick rasterizer check on the GPU box: surfel parity tests, then the bench line of both scenes (stage times, pairs, lane use)
# usage (through gpurun): bash tools/qb.sh <tag> [pytest -k expression]
tag=${1:-qb}
python -m pytest tests/test_surfel_gpu.py -m gpu -q -x ${2:+-k "$2"} 2>&1 | tail -4
for scene in surface stress; do
  python bench.py --no-dit --no-cpu-baseline --scene $scene > gpurun_out/${tag}_${scene}.json 2> gpurun_out/${tag}_${scene}.err || tail -5 gpurun_out/${tag}_${scene}.err
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_${scene}.json"))
print("${scene}", "ms/step", d["ms_per_step"], d["stage_ms"], "pairs", d["blend_valu"]["pairs_evaluated"], "lane use", d["blend_valu"]["lane_slot_utilisation"], "roofline", d["roofline"]["frac"])
PY
done
