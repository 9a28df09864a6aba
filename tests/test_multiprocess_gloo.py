"""world_size-2 and world_size-8 gloo runs (CPU) of the multi-GPU layout: sample sharding + the single gather to rank 0."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cascade_per_rank_with_stand_ins(gd, rank, world):
    """BASELINE configs[4] code path (distributed.cascade_per_rank, the one bench.py --gpus N runs): one cascaded sample per
    rank, own seed and conditioning, one gather to rank 0 -- with recording stand-ins for the two denoisers and the
    decoder (the HIP models need a GPU)."""
    class Den(torch.nn.Module):
        def __init__(self, C):
            super().__init__()
            self.in_channels = C
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward_with_cfg(self, x, t, context=None, cfg_scale=1.0):
            return -x + context["img_vector"].mean()

        def forward(self, x, t, context=None):
            return -x + context["img_vector"].mean()

    class Dec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.vit_decoder = torch.nn.Module()
            self.vit_decoder.pos_embed = torch.nn.Parameter(torch.zeros(1, 16, 4))

        def decode(self, latent, xyz):
            return {"latent": latent, "query_pcd_xyz": xyz}

        def triplane_decode(self, ret, cams, render_all_scale=True):
            v = float(ret["latent"].mean())
            mk = lambda ch: torch.full((1, 2, ch, 4, 4), v)  # noqa: E731
            return {"gaussians_upsampled_3": {"image": mk(3), "depth": mk(1), "alpha": mk(1), "rend_normal": mk(3), "dist": mk(1)}}

    def cond_fn(i):
        g = torch.Generator().manual_seed(100 + i)
        cond = {"img_crossattn": torch.randn(1, 5, 8, generator=g), "img_vector": torch.randn(1, 8, generator=g)}
        return cond, {k: torch.zeros_like(v) for k, v in cond.items()}

    gathered, mine = gd.cascade_per_rank(Den(3), Den(10), Dec(), cond_fn, {"tanfov": 0.36}, world, base_seed=7, num_steps=4,
                                         sampling_method="euler")
    if mine != [rank]:
        return False
    # two samples per rank: both rounds come back, in sample order (rank r owns samples 2r, 2r + 1), and a sample's render does
    # not depend on how the samples were sharded (sample `rank` of the one-per-rank run was seeded 7 + rank as well)
    g2, mine2 = gd.cascade_per_rank(Den(3), Den(10), Dec(), cond_fn, {"tanfov": 0.36}, 2 * world, base_seed=7, num_steps=4,
                                    sampling_method="euler")
    if mine2 != [2 * rank, 2 * rank + 1]:
        return False
    try:   # not a multiple of the world size: every rank refuses BEFORE any collective (nobody is left waiting in a gather)
        gd.cascade_per_rank(Den(3), Den(10), Dec(), cond_fn, {"tanfov": 0.36}, world + 1, base_seed=7, num_steps=4)
        return False
    except ValueError:
        pass
    if rank != 0:
        return gathered is None and g2 is None
    # every rank's payload arrived, and they differ (own seed, own conditioning)
    ok = tuple(gathered.shape) == (world, 2, 9, 4, 4) and float((gathered[0] - gathered[1]).abs().max()) > 0
    ok = ok and tuple(g2.shape) == (2 * world, 2, 9, 4, 4) and all(torch.equal(g2[i], gathered[i]) for i in range(world))
    return ok and len({float(g2[i].sum()) for i in range(2 * world)}) == 2 * world


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from gaussiananything_amd import distributed as gd
    r, w = gd.init_from_env(backend="gloo")
    mine = gd.shard_samples(5, r, w)
    color = torch.full((2, 3, 4, 4), float(r))
    allmap = torch.full((2, 7, 4, 4), float(10 + r))
    out = gd.gather_to_rank0(gd.pack_views(color, allmap))
    t = gd.max_over_ranks(1.0 + r, torch.device("cpu"))
    casc = _cascade_per_rank_with_stand_ins(gd, r, w)
    if r == 0:
        ok = out.shape == (w, 2, 10, 4, 4) and all(float(out[i, 0, 0, 0, 0]) == i and float(out[i, 0, 9, 0, 0]) == 10 + i
                                                    for i in range(w))
        q.put((ok and casc, mine, t))
    else:
        assert out is None
        q.put((casc, mine, t))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert sorted(sum((r[1] for r in res), [])) == list(range(5))      # shard_samples(5, ...): every sample exactly once
    assert all(abs(r[2] - float(world)) < 1e-9 for r in res)           # max over ranks of 1 + rank


def test_world_size_two_gloo():
    _run_world(2)


def test_world_size_eight_gloo():
    """the world size of BASELINE configs[4]: the [world, rounds, ...] -> sample-order reshuffle of cascade_per_rank and the gather
    with eight ranks (two sample rounds per rank)"""
    _run_world(8)


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (torch.distributed.run re-exec), rendezvous and
    report n_gpus = 2 -- never silently run one rank.  GA_BENCH_LAUNCH_ONLY=1 stops every rank after the rendezvous (gloo), so the
    launcher is exercised without a GPU; a WORLD_SIZE that contradicts --gpus is refused."""
    import json
    import subprocess
    env = dict(os.environ, GA_BENCH_LAUNCH_ONLY="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec == {"launch_only": True, "n_gpus": 2, "ranks_seen": 2}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in (bad.stderr + bad.stdout)
