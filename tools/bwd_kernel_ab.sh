#!/bin/bash
# per-kernel time of the backward for library variants (tools/bwd_variants.sh): rocprofv3 kernel trace of tools/bwd_bench.py.
# usage (GPU box): bash tools/bwd_kernel_ab.sh name1 name2 ...   (main = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
cd /tmp && export TMPDIR=/tmp
for name in "$@"; do
  if [ "$name" = main ]; then cp /tmp/main_backup.so $MAIN; else cp $R/tools/_build/libga_$name.so $MAIN; fi
  rm -rf /tmp/bk_$name
  rocprofv3 --kernel-trace --stats -d /tmp/bk_$name -o x -- python $R/tools/bwd_bench.py --reps 10 --scene ${SCENE:-surface} > /dev/null 2>/tmp/bk_$name.err
  echo "== $name"
  python $R/tools/rocpd_stats.py $(ls /tmp/bk_$name/*.db /tmp/bk_$name/*/*.db 2>/dev/null | head -1) | grep "surfel_bwd_grad\|surfel_bwd_sums" | python -c "import sys
for l in sys.stdin:
    f=l.split(' | '); print('  ', f[0][:40].ljust(40), f[3] if len(f)>3 else f)"
done
cp /tmp/main_backup.so $MAIN
