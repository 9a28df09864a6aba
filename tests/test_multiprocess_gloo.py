"""world_size-2 gloo run (CPU) of the multi-GPU layout: sample sharding + the single gather to rank 0."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from gaussiananything_amd import distributed as gd
    r, w = gd.init_from_env(backend="gloo")
    mine = gd.shard_samples(5, r, w)
    color = torch.full((2, 3, 4, 4), float(r))
    allmap = torch.full((2, 7, 4, 4), float(10 + r))
    out = gd.gather_to_rank0(gd.pack_views(color, allmap))
    t = gd.max_over_ranks(1.0 + r, torch.device("cpu"))
    if r == 0:
        ok = out.shape == (w, 2, 10, 4, 4) and all(float(out[i, 0, 0, 0, 0]) == i and float(out[i, 0, 9, 0, 0]) == 10 + i
                                                    for i in range(w))
        q.put((ok, mine, t))
    else:
        assert out is None
        q.put((True, mine, t))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_two_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert sorted(sum((r[1] for r in res), [])) == list(range(5))
    assert all(abs(r[2] - 2.0) < 1e-9 for r in res)
