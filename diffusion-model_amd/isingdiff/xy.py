"""2-D XY model on the GPU: the counterpart of the reference's C program ``XY_model/xymetoropolis.c``.

Angles are in turns (theta / 2 pi in [0, 1)), as in the reference's ``field2d``.  ``xy_sampling`` keeps the meaning
of the reference's function (``:146-245``): equilibrate, then average energy per site, magnetisation and the numbers
of +1 / -1 vortices over ``nsamples`` measurements.  The reference runs ONE chain, one random site per step
(10 000 sweeps of equilibration, a measurement every 10 sweeps); here many independent chains run side by side,
checkerboard sweeps with Philox uniforms (same stationary distribution), so the same number of measurements needs
far fewer sweeps per chain.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from . import _lib


def _check(theta: torch.Tensor):
    if not (isinstance(theta, torch.Tensor) and theta.is_cuda and theta.dtype == torch.float32 and theta.dim() == 3
            and theta.shape[1] == theta.shape[2] and theta.is_contiguous()):
        raise ValueError("theta must be a contiguous (B, L, L) float32 CUDA tensor of angles in turns")


def xy_initialize(n_chains: int, L: int, *, seed: int = 42, device: int = 0) -> torch.Tensor:
    """Uniform random angles in [0, 1) (``xy_initialize`` ``:124-143``), one lattice per chain."""
    g = torch.Generator(device=f"cuda:{device}")
    g.manual_seed(seed)
    return torch.rand((n_chains, L, L), generator=g, device=f"cuda:{device}", dtype=torch.float32)


def sweep(theta: torch.Tensor, beta: float, n_sweeps: int, *, delta: float = 0.1, seed: int = 42, first_global_index: int = 0,
          sweep0: int = 0) -> torch.Tensor:
    """In-place checkerboard Metropolis sweeps (the move of ``xy_sampling`` ``:169-211``, ``delta`` = its 0.1)."""
    _check(theta)
    dev = theta.device
    with torch.cuda.device(dev):
        _lib.check(_lib.load().idm_xy_metropolis(dev.index, theta.data_ptr(), theta.shape[0], theta.shape[1], float(beta), float(delta),
                                                 int(n_sweeps), seed & 0xFFFFFFFFFFFFFFFF, first_global_index, sweep0,
                                                 torch.cuda.current_stream(dev).cuda_stream))
    return theta


def observe(theta: torch.Tensor):
    """(energy per site, magnetisation, +1 vortices, -1 vortices) per lattice: ``xyenergy`` ``:32-50``, ``xymagnetization``
    ``:53-69``, ``xyvortex`` ``:71-121``; float64 / int32 CUDA tensors of length B."""
    _check(theta)
    dev, B = theta.device, theta.shape[0]
    e = torch.empty(B, dtype=torch.float64, device=dev)
    m = torch.empty(B, dtype=torch.float64, device=dev)
    vp = torch.empty(B, dtype=torch.int32, device=dev)
    vn = torch.empty(B, dtype=torch.int32, device=dev)
    if B:
        with torch.cuda.device(dev):
            _lib.check(_lib.load().idm_xy_observe(dev.index, theta.data_ptr(), B, theta.shape[1], e.data_ptr(), m.data_ptr(),
                                                  vp.data_ptr(), vn.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return e, m, vp, vn


@dataclasses.dataclass
class SamplingStats:          # the reference's ``sampling_stats`` (``:23-29``)
    energy: float
    magnetization: float
    vortex_positive: float
    vortex_negative: float


def xy_sampling(L: int, beta: float, nsamples: int, *, n_chains: int = 256, equilibration_sweeps: int = 10000,
                sample_interval: int = 10, seed: int = 42, device: int = 0, theta: torch.Tensor | None = None):
    """Averages over ``nsamples`` measurements (rounded up to a multiple of ``n_chains``), like ``xy_sampling``
    (``:146-245``); returns (SamplingStats, final angles (n_chains, L, L))."""
    if theta is None:
        theta = xy_initialize(n_chains, L, seed=seed, device=device)
    n_chains = theta.shape[0]
    sweep(theta, beta, equilibration_sweeps, seed=seed)
    done = equilibration_sweeps
    rounds = max(1, -(-int(nsamples) // n_chains))
    acc = np.zeros(4)
    for _ in range(rounds):
        e, m, vp, vn = observe(theta)
        acc += [float(e.mean()), float(m.mean()), float(vp.double().mean()), float(vn.double().mean())]
        sweep(theta, beta, sample_interval, seed=seed, sweep0=done)
        done += sample_interval
    acc /= rounds
    return SamplingStats(*acc), theta
