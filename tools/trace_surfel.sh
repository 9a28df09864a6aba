#!/bin/bash
# kernel-trace statistics of the rasterizer stages.  Usage: tools/trace_surfel.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-trace}
mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o x -- python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --steps 30 --warmup 10 > $R/gpurun_out/$tag/bench.json 2>$R/gpurun_out/$tag/err.txt
python $R/tools/rocpd_stats.py $(ls $R/gpurun_out/$tag/*.db $R/gpurun_out/$tag/*/*.db 2>/dev/null | head -1) | python -c "import sys
for l in sys.stdin:
    f=l.split(\" | \"); print(f[0][:50].ljust(50), *f[1:])" | head -12
