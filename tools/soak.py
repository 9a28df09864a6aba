"""Soak / reproducibility run of the whole cascaded sample (BASELINE configs[3]) on one GPU: N samples back to back, alternating two
conditionings and the two samplers, every repetition of a (conditioning, sampler) pair must be BIT-identical to its first run -- the
captured sampler steps are replayed on rewritten buffers, the cached K / V / pooled vectors are refreshed in place, the device dopri5
decides on a fixed-order error norm: nothing may depend on what ran before.  Prints peak memory and the time per sample.
usage (GPU box): python tools/soak.py [repetitions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaussiananything_amd import distributed as gd, synthetic

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cams = synthetic.eval_cameras(8)
c = {"cam_view": cams["cam_view"][None].to(dev), "cam_view_proj": cams["cam_view_proj"][None].to(dev),
     "cam_pos": cams["cam_pos"][None].to(dev), "tanfov": cams["tanfov"]}
m1, m2, dec = bench.build_cascade_models(dev)

def cond_fn_for(k):
    def cond_fn(i):
        g = torch.Generator().manual_seed(5000 + k)
        cond = {"img_crossattn": torch.randn(1, 1369, 1024, generator=g).to(dev), "img_vector": torch.randn(1, 1024, generator=g).to(dev)}
        return cond, {q: torch.zeros_like(v) for q, v in cond.items()}
    return cond_fn

first, ok, t_sum, n = {}, True, {"euler": 0.0, "dopri5": 0.0}, {"euler": 0, "dopri5": 0}
for rep in range(reps):
    for k in (0, 1):
        for method, steps in (("dopri5", 250), ("euler", 60)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gathered, _ = gd.cascade_per_rank(m1, m2, dec, cond_fn_for(k), c, 1, base_seed=42 + k, num_steps=steps, sampling_method=method,
                                              render_all_scale=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            key = (k, method)
            if key not in first:
                first[key] = gathered.clone()
            else:
                same = bool(torch.equal(gathered, first[key]))
                ok &= same
                t_sum[method] += dt; n[method] += 1
                if not same:
                    print("MISMATCH", key, rep, float((gathered - first[key]).abs().max()))
    print(f"rep {rep}: ok so far {ok}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
assert bool(torch.isfinite(first[(0, 'euler')]).all())
print({"bit_identical_repetitions": ok, "sec_per_sample_dopri5": round(t_sum["dopri5"] / max(n["dopri5"], 1), 4),
       "sec_per_sample_euler60": round(t_sum["euler"] / max(n["euler"], 1), 4), "samples": 4 * reps})
