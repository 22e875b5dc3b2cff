"""Spectral-domain post-processing for the FFT-UNet variant: the GPU counterpart of
``ising_ifft`` (b_fft.py:12-17) plus the sign projection that follows it
(b_diffusion_model_fft_unet_2d_ising.py:214-217)."""
from __future__ import annotations

import torch

from . import _lib


def irfft2_sign(spec: torch.Tensor, return_real: bool = False):
    """spec: (B, L, L//2+1, 2) float32 CUDA tensor holding (Re, Im) of an rfft2 spectrum.

    Returns ±1 spins ``(B, L, L)`` int8 = ``np.where(np.fft.irfft2(re + 1j*im, s=(L,L)) > 0, 1, -1)``
    (and the float64 transform itself when ``return_real``)."""
    if not (isinstance(spec, torch.Tensor) and spec.is_cuda and spec.dtype == torch.float32):
        raise ValueError("spec must be a float32 CUDA tensor")
    if spec.dim() != 4 or spec.shape[3] != 2 or spec.shape[2] != spec.shape[1] // 2 + 1:
        raise ValueError("spec must be (B, L, L//2+1, 2)")
    spec = spec.contiguous()
    B, L = spec.shape[0], spec.shape[1]
    spins = torch.empty((B, L, L), dtype=torch.int8, device=spec.device)
    real = torch.empty((B, L, L), dtype=torch.float64, device=spec.device) if return_real else None
    if B > 0:
        with torch.cuda.device(spec.device):
            _lib.check(_lib.load().idm_irfft2_sign(spec.device.index, spec.data_ptr(), B, L, spins.data_ptr(),
                                                   None if real is None else real.data_ptr(),
                                                   torch.cuda.current_stream(spec.device).cuda_stream))
    return (spins, real) if return_real else spins
