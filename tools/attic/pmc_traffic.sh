#!/bin/bash
# HBM-side traffic of the blend kernel (FETCH_SIZE, WRITE_SIZE: two counter-only passes) for a list of library variants.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
for name in "$@"; do
  if [ "$name" != main ]; then cp $R/tools/_build/libga_$name.so $MAIN; else cp /tmp/main_backup.so $MAIN; fi
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pt_$set
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d /tmp/pt_$set -o x -- python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --steps 5 --warmup 2 > /dev/null 2>/tmp/pt_$set.err
    echo -n "$name "; python $R/tools/rocpd_pmc.py $(ls /tmp/pt_$set/*/*.db /tmp/pt_$set/*.db 2>/dev/null | head -1) 2>&1 | grep -A2 "surfel_blend" | grep avg
  done
done
cp /tmp/main_backup.so $MAIN
