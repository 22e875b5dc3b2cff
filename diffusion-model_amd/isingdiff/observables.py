"""Ising observables on the GPU (b_pure_metoropolis_2d_ising.py:42-60) and their histograms."""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from . import _lib


@dataclasses.dataclass
class Observables:
    """Per-lattice integer sums and their histograms for ±1 spins on an L×L periodic lattice.

    M = Σ s,  S = Σ s·s_down + s·s_right (each bond once);  E = −J·S − h·M,  m = M / L².
    ``hist_M[k]`` counts lattices with M = 2k − L²; ``hist_S[k]`` counts S = 2k − 2L².
    ``moments`` = [count, ΣM, Σ|M|, ΣM², ΣS, ΣS²] (int64).
    """
    L: int
    M: torch.Tensor
    S: torch.Tensor
    hist_M: torch.Tensor
    hist_S: torch.Tensor
    moments: torch.Tensor

    def energy(self, J: float = 1.0, h: float = 0.0) -> np.ndarray:
        return -J * self.S.cpu().numpy().astype(np.float64) - h * self.M.cpu().numpy().astype(np.float64)

    def magnetization(self) -> np.ndarray:
        return self.M.cpu().numpy().astype(np.float64) / (self.L * self.L)

    def mean_energy_per_site(self, J: float = 1.0, h: float = 0.0) -> float:
        c, sm, _, _, ss, _ = (int(v) for v in self.moments.cpu().numpy())
        return (-J * ss - h * sm) / (c * self.L * self.L)


def observe(spins: torch.Tensor) -> Observables:
    """spins: (B, L, L) int8 CUDA tensor of ±1."""
    if not (isinstance(spins, torch.Tensor) and spins.is_cuda and spins.dtype == torch.int8):
        raise ValueError("spins must be an int8 CUDA tensor")
    if spins.dim() != 3 or spins.shape[1] != spins.shape[2]:
        raise ValueError("spins must be (B, L, L)")
    spins = spins.contiguous()
    B, L, _ = spins.shape
    dev = spins.device
    M = torch.empty(B, dtype=torch.int32, device=dev)
    S = torch.empty(B, dtype=torch.int32, device=dev)
    hist_M = torch.zeros(L * L + 1, dtype=torch.int64, device=dev)
    hist_S = torch.zeros(2 * L * L + 1, dtype=torch.int64, device=dev)
    moments = torch.zeros(6, dtype=torch.int64, device=dev)
    if B > 0:
        with torch.cuda.device(dev):
            _lib.check(_lib.load().idm_observe(dev.index, spins.data_ptr(), B, L, M.data_ptr(), S.data_ptr(),
                                               hist_M.data_ptr(), hist_S.data_ptr(), moments.data_ptr(),
                                               torch.cuda.current_stream(dev).cuda_stream))
    return Observables(L, M, S, hist_M, hist_S, moments)


def sign_project(x: torch.Tensor) -> torch.Tensor:
    """np.where(x > 0, 1, -1) on the GPU (b_diffusion_model_unet_2d_ising.py:211); x == 0 -> -1."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        raise ValueError("x must be a float32 CUDA tensor")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    if x.numel():
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().idm_sign_project(x.device.index, x.data_ptr(), out.data_ptr(), x.numel(),
                                                    torch.cuda.current_stream(x.device).cuda_stream))
    return out


def ising_energy(x_lst, J: float, h: float) -> np.ndarray:
    """``ising_energy(x_lst, J, h)`` (b_pure_metoropolis_2d_ising.py:42-53) for ±1 lattices (B,L,L)."""
    s = x_lst if isinstance(x_lst, torch.Tensor) else torch.as_tensor(np.asarray(x_lst))
    s = s.to(torch.device("cuda", torch.cuda.current_device()) if not s.is_cuda else s.device).to(torch.int8)
    return observe(s).energy(J, h)


def ising_magnetization(x_lst) -> np.ndarray:
    """``ising_magnetization(x_lst)`` (b_pure_metoropolis_2d_ising.py:56-60)."""
    s = x_lst if isinstance(x_lst, torch.Tensor) else torch.as_tensor(np.asarray(x_lst))
    s = s.to(torch.device("cuda", torch.cuda.current_device()) if not s.is_cuda else s.device).to(torch.int8)
    return observe(s).magnetization()
