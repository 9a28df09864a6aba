#!/bin/bash
# same-box A/B of the DiT evaluation time between library builds (tools/_build/libga_<name>.so, or `main`), alternating.
# usage (GPU box): bash tools/ab_lib_dit.sh nameA nameB [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN $MAIN.ab_backup
for r in $(seq 1 ${3:-2}); do
  for name in $1 $2; do
    if [ "$name" = main ]; then cp $MAIN.ab_backup $MAIN; else cp $R/tools/_build/libga_$name.so $MAIN; fi
    echo -n "$name: "; (cd $R && python tools/ab_dit.py child 2>/dev/null | tr '\n' ' '); echo
  done
done
cp $MAIN.ab_backup $MAIN; rm $MAIN.ab_backup
