#!/bin/bash
# same-box A/B between the in-tree library and the variant builds tools/_build/<name>/libga_mi355.so, alternating, two rounds:
# DiT ms per evaluation (L, B, L x 4 samples).  usage (GPU box): bash tools/ab_lib.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
for r in 1 2; do
  for name in main "$@"; do
    if [ "$name" = main ]; then cp /tmp/main_backup.so $MAIN;
    else cp $R/tools/_build/$name/libga_mi355.so $MAIN; fi
    echo -n "$name: "; (cd $R && python tools/ab_dit3.py child 2>/dev/null | tail -1)
  done
done
cp /tmp/main_backup.so $MAIN
