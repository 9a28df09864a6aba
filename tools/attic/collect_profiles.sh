#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/<tag>/ : kernel-trace stats of the rasterizer, PMC passes of
# the blend kernel (own runs, counters only), the default bench.py line.  Usage: tools/collect_profiles.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r1x}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
python bench.py > $out/bench.json 2> $out/bench.err
tools/trace_surfel.sh $tag/trace > $out/surfel_kernel_stats.txt 2>&1
python tools/rocpd_stats.py $(ls $out/trace/*.db | head -1) > $out/surfel_kernel_stats_full.txt
KPAT=surfel_blend tools/pmc_blend.sh $tag/pmc > $out/pmc.txt 2>&1
