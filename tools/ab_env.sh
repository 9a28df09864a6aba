#!/bin/bash
# same-box A/B of an environment switch, alternating, three rounds: DiT ms per evaluation (L, B, L x 4 samples).
# usage (GPU box): bash tools/ab_env.sh GA_DIT_TILED=0 GA_DIT_TILED=1
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for kv in "$@"; do
    echo -n "$kv: "; (cd $R && env $kv python tools/ab_dit3.py child 2>/dev/null | tail -1)
  done
done
