#!/bin/bash
# two PMC passes (instruction mix / waits, LDS) over the blend kernel for a list of library variants (tools/_build/libga_<name>.so)
cd /tmp && export TMPDIR=/tmp
R=${R_OVERRIDE:-${GRAFT_REPO_ROOT:-/root/repo}}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
for name in "$@"; do
  if [ "$name" != main ]; then cp $R/tools/_build/libga_$name.so $MAIN; else cp /tmp/main_backup.so $MAIN; fi
  echo "=== $name"
  i=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
             "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY"; do
    i=$((i+1))
    rm -rf /tmp/pq_$i
    rocprofv3 --kernel-trace --pmc $set -d /tmp/pq_$i -o x -- python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --steps 5 --warmup 2 ${SCENE:+--scene $SCENE} > /dev/null 2>/tmp/pq_$i.err
    python $R/tools/rocpd_pmc.py $(ls /tmp/pq_$i/*/*.db /tmp/pq_$i/*.db 2>/dev/null | head -1) 2>&1 | grep -A9 "surfel_blend" | grep -v surfel_blend
  done
done
cp /tmp/main_backup.so $MAIN
