#!/bin/bash
# Counter passes of the rasterizer's FRONT-END kernels at BASELINE configs[1] (counters only: --kernel-trace + --pmc, one set per run):
# is the preprocess bound by its ALUs or by its stores?  Usage: tools/pmc_pre.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmcpre}
out=$R/gpurun_out/$tag
mkdir -p $out
{
echo "# surfel_preprocess_kernel / surfel_fill_sched_kernel / surfel_run_sort_kernel at BASELINE configs[1] (python bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --no-extras --steps 5 --warmup 2)"
for SET in "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm_pre
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pm_pre -o x -- python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --no-extras --steps 5 --warmup 2 > /dev/null 2>/tmp/pm_pre.err
  python $R/tools/rocpd_pmc.py $(ls /tmp/pm_pre/*/*.db /tmp/pm_pre/*.db 2>/dev/null | head -1) 2>&1 | grep -A9 "surfel_preprocess\|surfel_fill_sched\|surfel_run_sort"
done
} > $out/frontend_pmc.txt 2>&1
cat $out/frontend_pmc.txt
