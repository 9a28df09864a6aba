#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-trd}
mkdir -p $R/gpurun_out/$tag
cat > /tmp/run_decode.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, bench, json
from gaussiananything_amd import synthetic
print(json.dumps(bench.bench_decode(torch.device("cuda:0"), synthetic.eval_cameras(8))))
PY
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o x -- python /tmp/run_decode.py > $R/gpurun_out/$tag/out.txt 2>$R/gpurun_out/$tag/err.txt
cat $R/gpurun_out/$tag/out.txt
python $R/tools/rocpd_stats.py $(ls $R/gpurun_out/$tag/*.db | head -1) | python -c "
import sys
for l in sys.stdin:
    f=l.split(' | ')
    print(f[0][:64].ljust(64), *f[1:4])" | head -24
