"""Ising configurations by checkerboard Metropolis on the GPU.

``pure_metropolis`` keeps the signature and return values of the reference's generator
(``b_pure_metoropolis_2d_ising.py:63-105``) and its acceptance rule (``:35-38``), but instead of one
long chain measured every ``measurement_interval_mcs`` sweeps it runs one independent chain per
requested sample: every chain starts from ``x_ini``, is equilibrated for ``n_mcs_equilibration``
sweeps and advanced ``measurement_interval_mcs`` more.  The samples are therefore uncorrelated.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def acceptance_table(beta: float, J: float, h: float) -> np.ndarray:
    """float32 exp(-beta * dE), dE = 2 s (J nn + h), rows s = -1, +1, columns nn = -4, -2, 0, 2, 4."""
    s = np.array([-1.0, 1.0])[:, None]
    nn = np.array([-4.0, -2.0, 0.0, 2.0, 4.0])[None, :]
    return np.ascontiguousarray(np.exp(-float(beta) * 2.0 * s * (float(J) * nn + float(h))).astype(np.float32))


def sweep(spins: torch.Tensor, beta: float, J: float, h: float, n_sweeps: int, *, seed: int = 42,
          first_global_index: int = 0, sweep0: int = 0) -> torch.Tensor:
    """In-place checkerboard sweeps of (B, L, L) int8 CUDA spins; returns ``spins``."""
    if not (spins.is_cuda and spins.dtype == torch.int8 and spins.dim() == 3 and spins.shape[1] == spins.shape[2]):
        raise ValueError("spins must be a (B, L, L) int8 CUDA tensor")
    if not spins.is_contiguous():
        raise ValueError("spins must be contiguous")
    tab = acceptance_table(beta, J, h)
    dev = spins.device
    with torch.cuda.device(dev):
        _lib.check(_lib.load().idm_metropolis(dev.index, spins.data_ptr(), spins.shape[0], spins.shape[1], tab.ctypes.data,
                                              int(n_sweeps), seed, first_global_index, sweep0,
                                              torch.cuda.current_stream(dev).cuda_stream))
    return spins


def pure_metropolis(x_ini, beta: float, J: float, h: float, n_mcs_equilibration: int, n_mcs_measurement: int,
                    measurement_interval_mcs: int, progress: bool = False, *, seed: int = 42, device: int = 0):
    """``pure_metropolis`` of the reference: returns (samples (n, L, L), energies (n,), magnetisations (n,)) with
    n = ceil(n_mcs_measurement / measurement_interval_mcs), as numpy arrays."""
    from .observables import observe
    x_ini = np.asarray(x_ini)
    L = x_ini.shape[0]
    n = -(-int(n_mcs_measurement) // int(measurement_interval_mcs))
    spins = torch.as_tensor(x_ini.astype(np.int8), device=f"cuda:{device}")[None].repeat(n, 1, 1).contiguous()
    sweep(spins, beta, J, h, int(n_mcs_equilibration) + int(measurement_interval_mcs), seed=seed)
    obs = observe(spins)
    energy = -float(J) * obs.S.cpu().numpy().astype(np.float64) - float(h) * obs.M.cpu().numpy().astype(np.float64)
    return spins.cpu().numpy().astype(x_ini.dtype if x_ini.dtype.kind in "if" else np.int8), energy, \
        obs.M.cpu().numpy().astype(np.float64) / (L * L)
