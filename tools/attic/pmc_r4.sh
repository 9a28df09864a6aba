#!/bin/bash
# Round-4 counter evidence (counters only: --kernel-trace + --pmc, one set per run).  Usage: tools/pmc_r4.sh <tag>
#   blend: instruction mix, LDS, FETCH_SIZE / WRITE_SIZE of surfel_blend_kernel at BASELINE configs[1];  attention: MFMA busy cycles of the
#   two bench shapes.  tools/pmc_to_json.py turns the two text files into the profiles/r4_*_pmc.json that bench.py quotes.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmc4}
out=$R/gpurun_out/$tag
mkdir -p $out
run() {  # name, kernel pattern, command..., counters in $SET
  local name=$1 pat=$2; shift 2
  rm -rf /tmp/pm_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm_$name -o x -- "$@" > /dev/null 2>/tmp/pm_$name.err
  python $R/tools/rocpd_pmc.py $(ls /tmp/pm_$name/*/*.db /tmp/pm_$name/*.db 2>/dev/null | head -1) 2>&1 | grep -A10 "$pat"
}
{
echo "# surfel_blend_kernel<false> at BASELINE configs[1] (python bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --steps 5 --warmup 2)"
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  run blend surfel_blend python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --steps 5 --warmup 2
done
} > $out/blend_pmc.txt 2>&1
{
echo "# attention_fwd_kernel, python tools/dit_kernels_two.py attn (20 launches of each bench shape: self 2x16x768x768, cross 1x16x768x1369)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  run attn attention_fwd python $R/tools/dit_kernels_two.py attn
done
} > $out/attention_pmc.txt 2>&1
wc -l $out/*.txt
