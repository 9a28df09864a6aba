#!/bin/bash
# blend-stage time (HIP events) and lane use of alternative builds (tools/blend_variants.sh).  usage (GPU box): bash tools/blend_ab_stage.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN $MAIN.ab_backup
for name in "$@"; do
  if [ "$name" = main ]; then cp $MAIN.ab_backup $MAIN; else cp $R/tools/_build/libga_$name.so $MAIN; fi
  for scene in surface stress; do
    python $R/bench.py --no-dit --no-cpu-baseline --no-parity --scene $scene 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name $scene', 'ms/step', d['ms_per_step'], 'blend', d['stage_ms']['blend'], 'iters', d['blend_valu'].get('wave_iterations'), 'lane use', d['blend_valu']['lane_slot_utilisation'])"
  done
done
cp $MAIN.ab_backup $MAIN; rm $MAIN.ab_backup
