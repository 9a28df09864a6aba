"""Multi-GPU layout of the hot path: independent samples shard one per rank with NO data-path
collective; the only exchange is the final collection of the rendered RGB-D-N images on rank 0.

The reference runs inference single-process (/root/reference/shell_scripts/release/inference/i23d/i23d-stage1.sh:15,143-146)
and its only process-group use is guided_diffusion/dist_util.py:57-75 (NCCL, env:// rendezvous); here one process per
GPU talks RCCL over xGMI through ``torch.distributed`` (backend "nccl" on ROCm), and the same code runs on ``gloo``/CPU
for the world_size-2 tests.  xGMI is point-to-point: every peer has its own link to rank 0, so a plain gather (7 concurrent
senders) is the right collective -- not a ring.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kw)
    return rank, world


def shard_samples(num_samples: int, rank: int, world: int):
    """Sample (seed) indices owned by ``rank``: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(num_samples, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def pack_views(color: torch.Tensor, allmap: torch.Tensor) -> torch.Tensor:
    """[V,3,H,W] + [V,7,H,W] -> one contiguous [V,10,H,W] payload (RGB, depth, alpha, normal xyz, median depth, dist)."""
    return torch.cat([color, allmap], dim=1).contiguous()


def pack_render(render: dict, item: int = 0) -> torch.Tensor:
    """One sample's rendered views of one level (the dict of ``GaussianRenderer2DGS.render``) -> contiguous [V,9,H,W]:
    RGB (3), median depth (1), alpha (1), world-space normal (3), distortion (1) -- the "multi-view RGB-D-N" payload."""
    return torch.cat([render["image"][item], render["depth"][item], render["alpha"][item], render["rend_normal"][item],
                      render["dist"][item]], dim=1).contiguous()


def cascade_per_rank(stage1, stage2, decoder, cond_fn, cameras, num_samples, base_seed=42, level="gaussians_upsampled_3",
                     **cascade_kwargs):
    """BASELINE configs[4]: ``num_samples`` independent cascaded samples sharded over the ranks (``shard_samples``), every
    rank running ``cascade.cascade`` for its own seeds with NO data-path collective; the rendered multi-view RGB-D-N of the
    finest level is collected on rank 0 with ONE gather per owned sample round.  ``cond_fn(sample_index) -> (cond, uc)``.
    ``num_samples`` must be a multiple of the world size.  Returns (gathered [num_samples, V, 9, H, W] in sample order on rank 0
    else None, list of this rank's sample indices)."""
    from . import cascade as _cascade
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if num_samples <= 0 or num_samples % world != 0:
        # checked on EVERY rank before any work: a rank that ran out of samples would otherwise leave the others waiting in
        # the gather of the last round
        raise ValueError(f"num_samples ({num_samples}) must be a positive multiple of the world size ({world})")
    mine = shard_samples(num_samples, rank, world)
    rounds = []
    for idx in mine:   # every rank owns num_samples / world samples: one gather per round, all rounds returned
        cond, uc = cond_fn(idx)
        out = _cascade.cascade(stage1, stage2, decoder, cond, uc, cameras=cameras, seed=base_seed + idx, **cascade_kwargs)
        rounds.append(gather_to_rank0(pack_render(out["renders"][level])))
    # rank 0: [world, rounds, V, 9, H, W] -> sample order (rank r owns the contiguous block r * rounds .. )
    gathered = torch.stack(rounds, dim=1).flatten(0, 1) if rounds[0] is not None else None
    return gathered, mine


def gather_to_rank0(payload: torch.Tensor, dst: int = 0):
    """Collect every rank's payload on ``dst``: returns [world, *payload.shape] there, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return payload.unsqueeze(0)
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == dst:
        out = torch.empty((world,) + tuple(payload.shape), dtype=payload.dtype, device=payload.device)
        dist.gather(payload, list(out.unbind(0)), dst=dst)
        return out
    dist.gather(payload, None, dst=dst)
    return None


def max_over_ranks(seconds: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
