#!/bin/bash
# backward check on the GPU box: parity tests, timing of both scenes, kernel-trace statistics.  Usage: bash tools/bwd_trace.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-bwd}
python -m pytest $R/tests/test_surfel_gpu.py -m gpu -q -x -k "backward or differentiable" 2>&1 | tail -3
python $R/tools/bwd_bench.py --scene surface | tee $R/gpurun_out/${tag}_surface.json
python $R/tools/bwd_bench.py --scene stress | tee $R/gpurun_out/${tag}_stress.json
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o x -- python $R/tools/bwd_bench.py --reps 10 --scene ${2:-surface} > /dev/null 2>$R/gpurun_out/$tag/err.txt
python $R/tools/rocpd_stats.py $(ls $R/gpurun_out/$tag/*.db $R/gpurun_out/$tag/*/*.db 2>/dev/null | head -1) | python -c "import sys
for l in sys.stdin:
    f=l.split(\" | \"); print(f[0][:50].ljust(50), *f[1:])" | head -40 | grep -v "^$" | tee $R/gpurun_out/${tag}_stats.txt
