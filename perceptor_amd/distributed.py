"""Replica sharding of independent sampling chains: one process per GPU, no collective on the data path.

perceptor is single-device; the chains of a batch are independent (GroupNorm, attention and
CLIP embeddings are per-sample), so rank r of R takes samples [r*N/R, (r+1)*N/R) of a batch whose initial
noise is drawn once on the CPU from a single seed (as the reference draws it: guided_diffusion.py:104), so the
INPUTS of every chain are invariant to R, and a chain's bits do not depend on which rank or batch position it runs
at (tested: batch permutation is bit-exact).  They do depend, at rounding level, on the per-rank batch SIZE (tile
and split-K choices follow M): under weak scaling (fixed chains per GPU, what bench.py runs) the results are
bit-identical for any R; re-sharding a fixed global batch changes them by 16-bit rounding noise, which the sign-like
guidance clamp (predictions.py:147-154) can turn into isolated +-2e-6*scale flips.  The only collective is one all-gather of the final images (RCCL over xGMI when
the backend is "nccl"; "gloo" in the CPU tests).  The CLIP loss is a mean over the GLOBAL batch
(losses/clip/clip.py:99): pass n_total to losses.*.loss_and_grad so a shard reproduces its slice of the
single-process gradient.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str = "nccl"):
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if backend == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)        # one process per GPU; RCCL binds its communicator to the current device
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split; the first n_total % world ranks get one extra sample."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi]


def gather_images(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather the per-rank final images into the full batch (every rank gets the result)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
