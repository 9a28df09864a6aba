#!/bin/bash
# blend counters at BASELINE configs[1] (counters only: --kernel-trace + --pmc, one set per run).  Usage: tools/pmc_blend_r3.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-pmcb}
out=$R/gpurun_out/$tag
mkdir -p $out
{
echo "# surfel_blend_kernel<false> at BASELINE configs[1] (python bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --steps 5 --warmup 2)"
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmb
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmb -o x -- python $R/bench.py --no-cpu-baseline --no-dit --no-stage-events --no-parity --steps 5 --warmup 2 > /dev/null 2>/tmp/pmb.err
  python $R/tools/rocpd_pmc.py $(ls /tmp/pmb/*/*.db /tmp/pmb/*.db 2>/dev/null | head -1) 2>&1 | grep -A10 "surfel_blend"
done
} > $out/blend_pmc.txt 2>&1
cat $out/blend_pmc.txt
