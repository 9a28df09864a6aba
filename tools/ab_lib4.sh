#!/bin/bash
# same-box A/B between the in-tree library (`main`) and variant builds tools/_build/libga_<name>.so, alternating, three rounds:
# DiT ms per evaluation at the four cascade shapes (tools/ab_dit4.py).  usage (GPU box): bash tools/ab_lib4.sh main name ...
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
for r in 1 2 3; do
  for name in "$@"; do
    if [ "$name" = main ]; then cp /tmp/main_backup.so $MAIN; else cp $R/tools/_build/libga_$name.so $MAIN; fi
    echo -n "$name: "; (cd $R && python tools/ab_dit4.py 2>/dev/null | tail -1)
  done
done
cp /tmp/main_backup.so $MAIN
