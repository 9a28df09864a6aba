#!/bin/bash
# Round-6 batch D: the GEMM counter pass again on the final dit_gemm.hip (the remainder-tile / three-slot instances and the row-sum stride
# changed the file; the released models' instances compile to the same code, the quotation is keyed by the source hash all the same),
# and the kernel-trace statistics of the XL evaluation at the final HEAD.  Then: copy into profiles/, python tools/pmc_to_json.py r6
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6d
mkdir -p $out
cd $R
export PYTHONPATH=$R
bash tools/trace_sampler.sh r6d/xl DiT-PixArt-PCD-CLAY-XL 12 euler > $out/dit_XL_kernel_stats.txt 2>&1; head -14 $out/dit_XL_kernel_stats.txt | cut -c1-150
python tools/attn_hd_bench.py > $out/attn_hd_bench.txt 2>&1; cat $out/attn_hd_bench.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, kernel pattern, command...; counters in $SET
  local name=$1 pat=$2; shift 2
  rm -rf /tmp/pm_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm_$name -o x -- "$@" > /dev/null 2>/tmp/pm_$name.err
  python $R/tools/rocpd_pmc.py $(ls /tmp/pm_$name/*/*.db /tmp/pm_$name/*.db 2>/dev/null | head -1) 2>&1 | grep -A10 "$pat"
}
{
echo "# round 6, final dit_gemm.hip, tools/collect_r6d.sh"
echo "# GEMM launches of a DiT-L block, python tools/dit_kernels_two.py gemm (20 launches each, cold weights: qkv 1536x3072x1024, fc1 1536x4096x1024,"
echo "# fc2 1536x1024x4096, proj 1536x1024x1024, cross-attention q 768x1024x1024)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  run gemm gemm_ python $R/tools/dit_kernels_two.py gemm
done
} > $out/gemm_pmc.txt 2>&1
head -30 $out/gemm_pmc.txt
rm -rf $out/xl
