"""Reverse-diffusion sampler: drop-in for ``diffusion_backward`` and the post-processing that
follows it in the reference script (b_diffusion_model_unet_2d_ising.py:81-98, :203-213).

The whole T-step loop runs inside libisingdiff.so (one FFI crossing per call).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Optional

import numpy as np
import torch

from . import _lib
from .model import ScoreModel, _precision_code, _stream_ptr
from .observables import Observables, observe


def beta_schedule(T: int = 1000) -> np.ndarray:
    """The reference's noise schedule, ``float32(np.linspace(1e-4, 0.02, T))`` (:168-172)."""
    return np.linspace(1e-4, 0.02, T).astype(np.float32)


UPDATE_RULES = {"score": _lib.IDM_UPDATE_SCORE, "ddpm": _lib.IDM_UPDATE_DDPM}


def alpha_cumprod(beta_lst) -> np.ndarray:
    """``float32(np.cumprod(1 - noise_beta))`` (:169-173).  The product is taken in float64 whatever ``beta_lst`` comes
    in and cast last, as the reference does with its float64 ``noise_beta``: given the float64 schedule this is the
    reference's array exactly; given float32 betas it is what the library computes when handed none."""
    return np.cumprod(1.0 - np.asarray(beta_lst, np.float64)).astype(np.float32)


def _run(model: ScoreModel, x: torch.Tensor, betas: np.ndarray, T: int, *, noise: Optional[torch.Tensor],
         traj: Optional[torch.Tensor], scores: Optional[torch.Tensor], spins: Optional[torch.Tensor],
         precision: int, seed: int, first_global_index: int, init_state: bool, add_noise: bool, chunk: int,
         update_rule: str = "score"):
    if update_rule not in UPDATE_RULES:
        raise ValueError(f"update_rule must be one of {sorted(UPDATE_RULES)}")
    acp = alpha_cumprod(betas) if update_rule == "ddpm" else None
    betas = np.ascontiguousarray(betas, np.float32)
    if betas.shape != (T,):
        raise ValueError(f"beta_lst must have T={T} entries")
    B, H, W, _ = x.shape
    cfg = _lib.SampleCfg()
    cfg.T = T
    cfg.precision = precision
    cfg.betas_host = betas.ctypes.data
    cfg.seed = seed & 0xFFFFFFFFFFFFFFFF
    cfg.first_global_index = first_global_index
    cfg.init_state = int(init_state)
    cfg.add_noise = int(add_noise)
    cfg.chunk = chunk
    cfg.update_rule = UPDATE_RULES[update_rule]
    cfg.alpha_cumprod_host = None if acp is None else acp.ctypes.data
    p = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(model.device):
        _lib.check(_lib.load().idm_sample(model.handle, C.byref(cfg), x.data_ptr(), p(noise), p(traj), p(scores),
                                          p(spins), B, H, W, _stream_ptr(model.device)))


def diffusion_backward(x_asym, beta_lst, score_model: ScoreModel, T: int, *, noise=None, return_trajectory=True,
                       precision=None, seed: int = 42, first_global_index: int = 0, return_scores=False,
                       update_rule: str = "score"):
    """``diffusion_backward(x_asym, beta_lst, score_model, T)`` of the reference (:81-98).

    Returns ``(T+1, B, H, W, C)`` with index 0 the final state, like the reference
    (``return_trajectory=False`` returns only the final ``(B,H,W,C)`` state).

    noise: ``None`` draws each step's standard normals in-kernel from Philox(seed); an array
    ``(T, B,H,W,C)`` indexed by t injects them (deterministic mode: step t adds
    ``noise[t] * sqrt(beta_t)``); ``False`` disables the noise term.
    update_rule: ``"score"`` is the update the reference runs (:96); ``"ddpm"`` the commented-out alternative (:95),
    ``x = (x - beta_t / sqrt(1 - alpha_cumprod[t]) * score) / sqrt(1 - beta_t) + noise``.
    Output type follows ``x_asym`` (numpy or torch).
    """
    prec = score_model.precision if precision is None else _precision_code(precision)
    is_torch = isinstance(x_asym, torch.Tensor)
    x = (x_asym.to(score_model.device, torch.float32) if is_torch
         else torch.from_numpy(np.ascontiguousarray(x_asym, np.float32)).to(score_model.device)).clone().contiguous()
    if x.dim() != 4 or x.shape[3] != score_model.channels:
        raise ValueError(f"x_asym must be (B,H,W,{score_model.channels})")
    if isinstance(beta_lst, torch.Tensor):
        beta_lst = beta_lst.detach().cpu().numpy()
    add_noise = noise is not False
    z = None
    if noise is not None and noise is not False:
        z = (noise.to(score_model.device, torch.float32) if isinstance(noise, torch.Tensor)
             else torch.from_numpy(np.ascontiguousarray(noise, np.float32)).to(score_model.device)).contiguous()
        if tuple(z.shape) != (T,) + tuple(x.shape):
            raise ValueError("noise must be (T, B,H,W,C)")
    traj = torch.empty((T + 1,) + tuple(x.shape), dtype=torch.float32, device=x.device) if return_trajectory else None
    scores = torch.empty((T,) + tuple(x.shape), dtype=torch.float32, device=x.device) if return_scores else None
    if x.shape[0] > 0:
        _run(score_model, x, np.asarray(beta_lst), T, noise=z, traj=traj, scores=scores, spins=None, precision=prec,
             seed=seed, first_global_index=first_global_index, init_state=False, add_noise=add_noise, chunk=0,
             update_rule=update_rule)
    out = traj if return_trajectory else x
    conv = (lambda t: t) if is_torch else (lambda t: t.cpu().numpy())
    return (conv(out), conv(scores)) if return_scores else conv(out)


@dataclasses.dataclass
class SampleResult:
    spins: torch.Tensor          # (n, H, W[, C]) int8 on the GPU
    observables: Optional[Observables]
    x_final: Optional[torch.Tensor]


def sample(model: ScoreModel, n: int, L: int, *, T: int = 1000, betas=None, seed: int = 42, precision=None,
           first_global_index: int = 0, chunk: int = 0, keep_state: bool = False, observables: bool = True,
           x_T=None, noise=None, width: Optional[int] = None, update_rule: str = "score") -> SampleResult:
    """Draw ``n`` L×L lattices: x_T ~ N(0,1) (Philox, keyed by the global lattice index), the full
    reverse schedule, sign projection (:203-213) and — for single-channel models — the Ising
    observables.  Everything stays on the GPU."""
    prec = model.precision if precision is None else _precision_code(precision)
    betas = beta_schedule(T) if betas is None else np.asarray(betas, np.float32)
    # spectral models (C = 2) diffuse the rfft2 spectrum: state (n, L, L//2+1, 2)
    if width is None:
        width = L // 2 + 1 if model.channels == 2 else L
    shape = (n, L, width, model.channels)
    if x_T is None:
        x = torch.empty(shape, dtype=torch.float32, device=model.device)
        init = True
    else:
        x = torch.as_tensor(x_T, dtype=torch.float32).to(model.device).clone().contiguous()
        if tuple(x.shape) != shape:
            raise ValueError(f"x_T must have shape {shape}, got {tuple(x.shape)}")
        init = False
    z = None if noise is None else torch.as_tensor(noise, dtype=torch.float32).to(model.device).contiguous()
    if z is not None and tuple(z.shape) != (T,) + shape:
        raise ValueError(f"noise must have shape {(T,) + shape}, got {tuple(z.shape)}")
    spins = torch.empty(shape, dtype=torch.int8, device=model.device)
    if n > 0:
        _run(model, x, betas, T, noise=z, traj=None, scores=None, spins=spins, precision=prec, seed=seed,
             first_global_index=first_global_index, init_state=init, add_noise=True, chunk=chunk, update_rule=update_rule)
    obs = None
    if model.channels == 2 and width == L // 2 + 1 and n > 0:
        from .spectral import irfft2_sign
        spins = irfft2_sign(x)          # b_diffusion_model_fft_unet_2d_ising.py:214-217
    elif model.channels == 1 and width == L:
        spins = spins.reshape(n, L, L)
    if observables and n > 0 and spins.dim() == 3:
        obs = observe(spins)
    return SampleResult(spins, obs, x if keep_state else None)
