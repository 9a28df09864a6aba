#!/bin/bash
# Round-3 counter evidence (counters only: --kernel-trace + --pmc, one set per run).  Usage: tools/pmc_r3.sh <tag>
#   attention: MFMA busy cycles of the two bench shapes;  GEMM: MFMA ops / LDS / waits of the ring kernel at the DiT-L shapes;
#   blend: instruction mix, LDS, FETCH_SIZE / WRITE_SIZE at BASELINE configs[1]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmc3}
out=$R/gpurun_out/$tag
mkdir -p $out
run() {  # name, kernel pattern, command..., counters in $SET
  local name=$1 pat=$2; shift 2
  rm -rf /tmp/pm_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm_$name -o x -- "$@" > /dev/null 2>/tmp/pm_$name.err
  python $R/tools/rocpd_pmc.py $(ls /tmp/pm_$name/*/*.db /tmp/pm_$name/*.db 2>/dev/null | head -1) 2>&1 | grep -A10 "$pat"
}
{
echo "# attention_fwd_kernel, python tools/dit_kernels_two.py attn (20 launches of each bench shape; <8,1> = self 2x16x768x768, <4,2> = cross 1x16x768x1369)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  run attn attention_fwd python $R/tools/dit_kernels_two.py attn
done
} > $out/attention_pmc.txt 2>&1
{
echo "# gemm_ring_kernel, python tools/dit_kernels_two.py gemm (20 launches per shape, cold weights): <0,4,2,3,4,4,2> qkv, <1,4,2,...> fc1, <2,2,2,3,2,4,0> fc2 + proj, <0,4,1,1,4,4,0> cross-attention q (M = 768)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  run gemm gemm_ring python $R/tools/dit_kernels_two.py gemm
done
} > $out/gemm_pmc.txt 2>&1
{
echo "# surfel_blend_kernel at BASELINE configs[1] (python bench.py --no-cpu-baseline --no-dit --no-stage-events --steps 5 --warmup 2)"
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  run blend surfel_blend python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --steps 5 --warmup 2
done
} > $out/blend_pmc.txt 2>&1
wc -l $out/*.txt
