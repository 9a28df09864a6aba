"""stdin: one bench.py JSON line -> the DiT ms_per_nfe numbers on one line (A/B loops on a GPU box, e.g.
`for f in 0 1; do GA_DIT_FOLD_MOD=$f python bench.py --no-cpu-baseline --no-cascade --no-extras --no-parity --no-stage-events | python tools/_pick_dit.py; done`)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print([(x["arch"][-1], x["cfg_batch"], x["ms_per_nfe"]) for x in d["dit"]], "batched", d["dit_batched"]["ms_per_nfe"])
