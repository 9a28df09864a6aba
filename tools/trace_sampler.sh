#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-trs}
mkdir -p $R/gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o x -- python $R/tools/sampler_profile.py ${2:-DiT-PixArt-PCD-CLAY-L} ${3:-30} ${4:-euler} > $R/gpurun_out/$tag/out.txt 2>$R/gpurun_out/$tag/err.txt
cat $R/gpurun_out/$tag/out.txt
python $R/tools/rocpd_stats.py $(ls $R/gpurun_out/$tag/*.db | head -1) | python -c "
import sys
tot=0
for l in sys.stdin:
    f=l.split(' | ')
    if len(f)>3 and f[1].isdigit(): tot+=float(f[2])
    print(f[0][:64].ljust(64), *f[1:4])
print('TOTAL kernel us', tot)" | head -30
