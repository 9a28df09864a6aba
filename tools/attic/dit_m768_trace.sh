#!/bin/bash
# per-kernel split of the batch-1 stage-2 evaluation (tools/dit_m768.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/dit_m768.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/m768
rocprofv3 --kernel-trace --stats -d /tmp/m768 -o x -- python $R/tools/dit_m768.py DiT-PixArt-PCD-CLAY-stage2-L 20 > /dev/null 2>/tmp/m768.err
python $R/tools/rocpd_stats.py $(ls /tmp/m768/*.db /tmp/m768/*/*.db 2>/dev/null | head -1) | python -c "import sys
for l in sys.stdin:
    f=l.split(' | '); print(f[0][:72].ljust(72), *f[1:])" | head -16
