#!/bin/bash
# same-box A/B of an environment switch on the rasterizer forward (BASELINE configs[1] + the stress scene), alternating, three rounds.
# usage (GPU box): bash tools/ab_env_surfel.sh GA_SURFEL_FLAGS=8 GA_SURFEL_FLAGS=0
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for kv in "$@"; do
    echo -n "$kv: "; (cd $R && env $kv python bench.py --no-cpu-baseline --no-dit --no-parity --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('ms_step %.4f  Gsplats/s %.3f  pre %.1f fill %.1f sort %.1f blend %.1f  stress_ms %.4f' % (d['ms_per_step'], d['value']/1e3, s['preprocess']*1e3, s['tile_scan_fill']*1e3, s['tile_sort']*1e3, s['blend']*1e3, d['stress_scene']['ms_per_step']))")
  done
done
