import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print([(x['arch'][-1], x['cfg_batch'], x['ms_per_nfe']) for x in d['dit']], 'batched', d['dit_batched']['ms_per_nfe'])
