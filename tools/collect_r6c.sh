#!/bin/bash
# Round-6 batch C: split-K correctness re-test (compiler-visible hand-over), render_levels test, kernel-trace statistics of the XL evaluation,
# of the batch-1 (M = 768) stage-2 evaluation and of the rasterizer backward
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r6c
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_dit_gpu.py -x -q -rs -k "split_k or gemm" > $out/pytest_gemm.txt 2>&1; tail -4 $out/pytest_gemm.txt
GA_GEMM_SPLITK=6 timeout 600 python -m pytest tests/test_dit_gpu.py -x -q -k "golden or release_shape or reproducible" > $out/pytest_dit_sk6.txt 2>&1; tail -3 $out/pytest_dit_sk6.txt
timeout 300 python tools/splitk_bench.py > $out/splitk_bench2.txt 2>&1; cat $out/splitk_bench2.txt
bash tools/trace_sampler.sh r6c/xl DiT-PixArt-PCD-CLAY-XL 12 euler > $out/dit_XL_kernel_stats.txt 2>&1; head -40 $out/dit_XL_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $out/b1; rocprofv3 --kernel-trace --stats -d $out/b1 -o x -- python $R/tools/dit_m768.py DiT-PixArt-PCD-CLAY-stage2-L 20 > $out/b1_out.txt 2>/dev/null
{ cat $out/b1_out.txt; python $R/tools/rocpd_stats.py $(ls $out/b1/*.db $out/b1/*/*.db 2>/dev/null | head -1) | head -24; } > $out/dit_L_batch1_kernel_stats.txt; cut -c1-150 $out/dit_L_batch1_kernel_stats.txt
rm -rf $out/bw; rocprofv3 --kernel-trace --stats -d $out/bw -o x -- python $R/tools/bwd_bench.py > $out/bw_out.txt 2>/dev/null
{ cat $out/bw_out.txt; python $R/tools/rocpd_stats.py $(ls $out/bw/*.db $out/bw/*/*.db 2>/dev/null | head -1) | head -24; } > $out/backward_kernel_stats.txt; cut -c1-150 $out/backward_kernel_stats.txt
rm -rf $out/xl $out/b1 $out/bw
