#!/bin/bash
# Round-6 evidence, one GPU call: the default bench line, kernel-trace statistics (rasterizer forward, the two samplers), the counter
# passes (counters only: --kernel-trace + --pmc, one set per run) of the blend, the attention launches and the GEMM launches, and one
# world-size-1 run of the multi-rank code path (RCCL init, gather).  Usage: tools/collect_r6.sh <tag>;  then copy into profiles/ and run
# python tools/pmc_to_json.py
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r6}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
bash tools/trace_surfel.sh $tag/surfel_trace > $out/surfel_kernel_stats.txt 2>&1
bash tools/trace_sampler.sh $tag/sampler_euler DiT-PixArt-PCD-CLAY-L 30 euler > $out/dit_L_euler_kernel_stats.txt 2>&1
bash tools/trace_sampler.sh $tag/sampler_dopri5 DiT-PixArt-PCD-CLAY-L 250 dopri5 > $out/dit_L_dopri5_kernel_stats.txt 2>&1
cd /tmp && export TMPDIR=/tmp
run() {  # name, kernel pattern, command...; counters in $SET
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
echo "# attention_fwd_kernel, python tools/dit_kernels_two.py attn (20 launches each: self 2x16x768x768 <8,2>, cross 1x16x768x1369 <4,3>, and the same"
echo "# cross-attention with the q projection inside the workgroups -- the <4,3> block averages the two cross-attention forms)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  run attn attention_fwd python $R/tools/dit_kernels_two.py attn
done
} > $out/attention_pmc.txt 2>&1
{
echo "# GEMM launches of a DiT-L block, python tools/dit_kernels_two.py gemm (20 launches each, cold weights: qkv 1536x3072x1024, fc1 1536x4096x1024,"
echo "# fc2 1536x1024x4096, proj 1536x1024x1024, cross-attention q 768x1024x1024)"
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  run gemm gemm_ python $R/tools/dit_kernels_two.py gemm
done
} > $out/gemm_pmc.txt 2>&1
# the multi-rank code path at world size 1: RCCL process group, the gather of the rendered views inside the timed cascade sample
cd $R
{
echo "# GA_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 python bench.py --no-cpu-baseline --no-parity --no-extras under rocprofv3 --kernel-trace --stats:"
echo "# init_process_group('nccl') = RCCL, dist.gather of the [8,10,512,512] payload after the timed steps and of [8,9,512,512] inside cascade_per_rank"
cd /tmp
rm -rf /tmp/tr_dist
GA_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 \
  rocprofv3 --kernel-trace --stats -d /tmp/tr_dist -o x -- python $R/bench.py --no-cpu-baseline --no-parity --no-extras > $out/dist_bench.json 2> $out/dist_bench.err
python $R/tools/rocpd_stats.py $(ls /tmp/tr_dist/*.db /tmp/tr_dist/*/*.db 2>/dev/null | head -1) | grep -i "rccl\|nccl\|ncclDevKernel\|copyBuffer\|Generic" | head -12
python -c "import json,sys; d=json.loads(open('$out/dist_bench.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('n_gpus','rccl_ranks','value','sec_per_sample')}, d['config'].get('gather_ms'), d.get('cascade',{}).get('gathered_shape'))"
} > $out/dist_world1.txt 2>&1
ls -la $out
