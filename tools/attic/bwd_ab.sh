#!/bin/bash
# Time the backward with alternative builds of the library (tools/bwd_variants.sh).  usage (GPU box): bash tools/bwd_ab.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN $MAIN.ab_backup
for name in "$@"; do
  if [ "$name" = main ]; then cp $MAIN.ab_backup $MAIN; else cp $R/tools/_build/libga_$name.so $MAIN; fi
  for scene in surface stress; do
    echo -n "$name $scene: "; python $R/tools/bwd_bench.py --scene $scene --reps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['forward_ms_median'], d['loss_plus_backward_ms_median'], [round(x,1) for x in d['grad_norms']])"
  done
done
cp $MAIN.ab_backup $MAIN; rm $MAIN.ab_backup
