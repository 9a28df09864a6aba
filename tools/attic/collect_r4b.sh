#!/bin/bash
# Round-4 evidence after the DiT pre-norm folds (the rasterizer kernels and their counter passes are unchanged since tools/collect_r4.sh):
# the default bench line, kernel-trace statistics of the two samplers, the counter pass of the attention launches.  Usage: tools/collect_r4b.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r4y}
out=$R/gpurun_out/$tag
mkdir -p $out/pmc
cd $R
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
bash tools/trace_sampler.sh $tag/sampler_euler DiT-PixArt-PCD-CLAY-L 30 euler > $out/dit_L_euler_kernel_stats.txt 2>&1
bash tools/trace_sampler.sh $tag/sampler_dopri5 DiT-PixArt-PCD-CLAY-L 250 dopri5 > $out/dit_L_dopri5_kernel_stats.txt 2>&1
cd /tmp && export TMPDIR=/tmp
{
echo "# attention_fwd_kernel, python tools/dit_kernels_two.py attn (20 launches of each bench shape: self 2x16x768x768, cross 1x16x768x1369)"
rm -rf /tmp/pm_attn
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD -d /tmp/pm_attn -o x -- python $R/tools/dit_kernels_two.py attn > /dev/null 2>/tmp/pm_attn.err
python $R/tools/rocpd_pmc.py $(ls /tmp/pm_attn/*/*.db /tmp/pm_attn/*.db 2>/dev/null | head -1) 2>&1 | grep -A10 attention_fwd
} > $out/pmc/attention_pmc.txt 2>&1
ls -la $out $out/pmc
