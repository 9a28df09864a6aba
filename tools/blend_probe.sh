R=${GRAFT_REPO_ROOT:-/root/repo}
MAIN=$R/gaussiananything_amd/lib/libga_mi355.so
cp $MAIN /tmp/main_backup.so
for name in main probe1 probe2; do
  if [ "$name" = main ]; then cp /tmp/main_backup.so $MAIN; else cp $R/tools/_build/libga_$name.so $MAIN; fi
  echo -n "$name: "; (cd $R && python bench.py --no-cpu-baseline --no-dit --no-parity --no-extras --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('ms_step %.4f  pre %.1f fill %.1f sort %.1f blend %.1f' % (d['ms_per_step'], s['preprocess']*1e3, s['tile_scan_fill']*1e3, s['tile_sort']*1e3, s['blend']*1e3))")
done
cp /tmp/main_backup.so $MAIN
