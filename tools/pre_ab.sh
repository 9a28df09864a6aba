#!/bin/bash
# same-box A/B of the preprocess kernel's views-per-thread (GA_PRE_VIEWS), alternating: stage times of the bench forward.
# usage (GPU box): bash tools/pre_ab.sh 1:2 2:2 4:2 4:4   (views per thread : groups of 256 Gaussians per workgroup)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in 1 2; do
  for kv in "$@"; do
    vg=${kv%%:*}; sp=${kv##*:}
    echo -n "GA_PRE_VIEWS=$vg GA_PRE_SPLATS=$sp: "
    GA_PRE_NT=${NT:-0} GA_PRE_VIEWS=$vg GA_PRE_SPLATS=$sp python bench.py --no-dit --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
  done
done
