"""Multi-GPU sharding: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Lattices are independent for the whole schedule, so the batch is cut into contiguous blocks
of global indices, one per rank, with no data-path collective.  RNG counters use the global
lattice index, so the spins do not depend on the number of ranks.  The only exchange is one
all-reduce (sum, int64) of the observable histograms and moments at the end.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

from .observables import Observables


def shard_range(n_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [first, first+count) of rank ``rank``; earlier ranks take the remainder."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    q, r = divmod(n_total, world_size)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def allreduce_observables(obs: Observables, group=None) -> Observables:
    """Sum histograms and moments over all ranks (one fused int64 all-reduce).  Per-lattice
    M and S stay local to the rank that produced them."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return obs
    nm, ns = obs.hist_M.numel(), obs.hist_S.numel()
    flat = torch.cat([obs.hist_M, obs.hist_S, obs.moments])
    if dist.get_backend(group) == "gloo" and flat.is_cuda:
        # a host-side group (tests, rehearsals): reduce through host memory
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return Observables(obs.L, obs.M, obs.S, flat[:nm].clone(), flat[nm:nm + ns].clone(), flat[nm + ns:].clone())
