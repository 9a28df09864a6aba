#!/bin/bash
# Round-6 batch E: the attention counter pass again on the final dit_attention.hip (DMA source addresses in SADDR form) and the kernel-trace
# statistics of the XL evaluation at the final HEAD (tails behind both attention grids).  Then: copy into profiles/, python tools/pmc_to_json.py r6
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6e
mkdir -p $out
cd $R
export PYTHONPATH=$R
bash tools/trace_sampler.sh r6e/xl DiT-PixArt-PCD-CLAY-XL 12 euler > $out/dit_XL_kernel_stats.txt 2>&1; head -14 $out/dit_XL_kernel_stats.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp
run() {  # name, kernel pattern, command...; counters in $SET
  local name=$1 pat=$2; shift 2
  rm -rf /tmp/pm_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm_$name -o x -- "$@" > /dev/null 2>/tmp/pm_$name.err
  python $R/tools/rocpd_pmc.py $(ls /tmp/pm_$name/*/*.db /tmp/pm_$name/*.db 2>/dev/null | head -1) 2>&1 | grep -A10 "$pat"
}
{
echo "# round 6, final dit_attention.hip, tools/collect_r6e.sh"
echo "# attention_fwd_kernel, python tools/dit_kernels_two.py attn (20 launches each: self 2x16x768x768 <8,2>, cross 1x16x768x1369 <4,3>, and the same"
echo "# cross-attention with the q projection inside the workgroups -- the <4,3> block averages the two cross-attention forms)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  run attn attention_fwd python $R/tools/dit_kernels_two.py attn
done
} > $out/attention_pmc.txt 2>&1
cat $out/attention_pmc.txt | head -40
rm -rf $out/xl
