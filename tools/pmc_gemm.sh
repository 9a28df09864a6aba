#!/bin/bash
# PMC passes over one GEMM shape (counters only).  Usage: tools/pmc_gemm.sh <tag>   (env GA_GEMM_CFG selects the variant)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmcg}
mkdir -p $R/gpurun_out/$tag
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/$tag/p$i -o x -- python $R/tools/gemm_sweep.py > /dev/null 2>$R/gpurun_out/$tag/p$i.err
  python $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/$tag/p$i/*.db 2>/dev/null | head -1) 2>&1 | grep -A9 "gemm_bf16" | head -60
done
