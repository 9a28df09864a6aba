#!/bin/bash
# same-box A/B between the in-tree library and variant builds tools/_build/<name>/libga_mi355.so on the rasterizer: the parity tests of
# BASELINE configs[1] + the merge path, then the bench forward's stage times, alternating twice.  usage (GPU box): bash tools/ab_lib_surfel.sh name1 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
cd $R
for name in "$@"; do
  cp $R/tools/_build/$name/libga_mi355.so $MAIN
  echo "== $name: parity"; python -m pytest tests/test_surfel_gpu.py -m gpu -q -x -k "baseline_config2 or merge_path or segmented or ragged or graph_replay or 7681" 2>&1 | tail -2
done
for r in 1 2; do
  for name in main "$@"; do
    if [ "$name" = main ]; then cp /tmp/main_backup.so $MAIN; else cp $R/tools/_build/$name/libga_mi355.so $MAIN; fi
    for scene in surface stress; do
      echo -n "$name $scene: "; python bench.py --scene $scene --no-dit --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
    done
  done
done
cp /tmp/main_backup.so $MAIN
