#!/bin/bash
# PMC passes over the rasterizer (counters only, kernel-trace; no other trace domains).  Usage: tools/pmc_blend.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmc}
mkdir -p $R/gpurun_out/$tag
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/$tag/p$i -o x -- python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --steps 5 --warmup 2 > /dev/null 2>$R/gpurun_out/$tag/p$i.err
  python $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/$tag/p$i/*/*.db $R/gpurun_out/$tag/p$i/*.db 2>/dev/null | head -1) 2>&1 | grep -A12 "${KPAT:-surfel_blend}"
done
