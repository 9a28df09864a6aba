#!/bin/bash
# Round-4 evidence, one GPU call: the default bench line, kernel-trace statistics of the rasterizer forward and of the two samplers
# (Euler replay, device dopri5), the counter passes of the blend and of the attention launches.  Usage: tools/collect_r4.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r4x}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
bash tools/trace_surfel.sh $tag/trace > $out/surfel_kernel_stats.txt 2>&1
python tools/rocpd_stats.py $(ls $out/trace/*.db | head -1) > $out/surfel_kernel_stats_full.txt
bash tools/trace_sampler.sh $tag/sampler_euler DiT-PixArt-PCD-CLAY-L 30 euler > $out/dit_L_euler_kernel_stats.txt 2>&1
bash tools/trace_sampler.sh $tag/sampler_dopri5 DiT-PixArt-PCD-CLAY-L 250 dopri5 > $out/dit_L_dopri5_kernel_stats.txt 2>&1
bash tools/pmc_r4.sh $tag/pmc > $out/pmc.log 2>&1
ls -la $out $out/pmc
