#!/bin/bash
# same-box A/B of environment switches, alternating, three rounds: DiT ms per evaluation at the four cascade shapes (tools/ab_dit4.py).
# usage (GPU box): bash tools/ab_env4.sh GA_GEMM_SPLITK=0 GA_GEMM_SPLITK=-1
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for kv in "$@"; do
    echo -n "$kv: "; (cd $R && env $kv python tools/ab_dit4.py 2>/dev/null | tail -1)
  done
done
