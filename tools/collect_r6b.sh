#!/bin/bash
# Round-6 measurement batch B: split-K tests + yardstick + same-box DiT A/B, render_levels test + video-path A/B, cascade parity chain
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6b
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_dit_gpu.py -x -q -rs -k "split_k or gemm or final_layer or golden or release" > $out/pytest_dit.txt 2>&1; tail -4 $out/pytest_dit.txt
timeout 600 python -m pytest tests/test_surfel_gpu.py tests/test_decode_gpu.py -x -q -rs -k "render_levels or triplane or renderer" > $out/pytest_levels.txt 2>&1; tail -4 $out/pytest_levels.txt
timeout 600 python tools/splitk_bench.py > $out/splitk_bench.txt 2>&1; cat $out/splitk_bench.txt
timeout 900 bash tools/ab_env4.sh GA_GEMM_SPLITK=0 GA_GEMM_SPLITK=-1 GA_GEMM_SPLITK=2 > $out/ab_splitk.txt 2>&1; cat $out/ab_splitk.txt
for r in 1 2 3; do for kv in GA_RENDER_LEVELS=0 GA_RENDER_LEVELS=1; do echo -n "$kv: "; env $kv python - <<'PY' 2>/dev/null | tail -1
import torch, bench
from gaussiananything_amd import synthetic
d = bench.bench_decode(torch.device("cuda:0"), synthetic.eval_cameras(8), reps=3)
print(d["video_50views_x_4levels_ms"], d["raster_8x512_ms"], d["ms_per_decode"])
PY
done; done > $out/ab_levels.txt 2>&1; cat $out/ab_levels.txt
timeout 1200 python tools/parity_r6.py cascade > $out/cascade25.txt 2> $out/cascade25.err; tail -30 $out/cascade25.txt
