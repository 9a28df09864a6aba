#!/bin/bash
# PMC passes over the attention kernel (counters only).  Usage: tools/pmc_attn.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmca}
mkdir -p $R/gpurun_out/$tag
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/$tag/p$i -o x -- python $R/tools/attn_bench.py > $R/gpurun_out/$tag/out$i.txt 2>$R/gpurun_out/$tag/p$i.err
  python $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/$tag/p$i/*.db 2>/dev/null | head -1) 2>&1 | grep -A9 "attention_fwd"
done
cat $R/gpurun_out/$tag/out1.txt
