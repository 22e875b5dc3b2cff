"""TEST INFRASTRUCTURE — PyTorch-CPU restatement of the reference's Keras sampling path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does.

PARITY UNPINNED by the reference's own tests: the reference has no tests, golden files
or recorded outputs for this path, and Keras 3.10 / TensorFlow 2.19 (its arithmetic
backend) are not installed here.  What pins this restatement instead: the saved weights,
``config.json`` inside each ``.keras`` file, the trained-objective known answers checked in
``tests/test_oracle.py`` (the model must predict the injected noise), and agreement with
the independent C restatement in ``oracle/ref_cpu.c``.

Follows:
  create_unet_model        b_diffusion_model_unet_2d_ising.py:22-78
                           b_diffusion_model_fft_unet_2d_ising.py:25-92 (BatchNorm variant)
  diffusion_backward       b_diffusion_model_unet_2d_ising.py:81-98
  sign projection          b_diffusion_model_unet_2d_ising.py:209-213
  ising_energy / magnetization   b_pure_metoropolis_2d_ising.py:42-60
Uses library convolutions (oneDNN), so its summation order differs from the C oracle:
the two agree to rounding, not bit for bit.  This is also the CPU baseline timed by
bench.py, because library convolutions are what the Keras CPU path runs on.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class TorchUNet:
    """Functional UNet over NCHW tensors; weights given as a ``UNetSpec``-like object."""

    def __init__(self, spec, dtype=torch.float32):
        self.dtype = dtype
        self.channels = spec.channels
        self.convs = []
        for c in spec.convs:
            w = torch.from_numpy(np.ascontiguousarray(c.kernel.transpose(3, 2, 0, 1))).to(dtype)
            b = torch.from_numpy(c.bias).to(dtype)
            sc = None if c.bn_scale is None else torch.from_numpy(c.bn_scale).to(dtype)
            sh = None if c.bn_shift is None else torch.from_numpy(c.bn_shift).to(dtype)
            self.convs.append((w, b, sc, sh, c.silu))
        self.dense = [(torch.from_numpy(d.kernel).to(dtype), torch.from_numpy(d.bias).to(dtype))
                      for d in spec.dense]

    def _conv(self, i, x):
        w, b, sc, sh, silu = self.convs[i]
        y = F.conv2d(x, w, b, padding=w.shape[-1] // 2)
        if sc is not None:
            y = y * sc[None, :, None, None] + sh[None, :, None, None]
        return F.silu(y) if silu else y

    def time_add(self, t_norm: torch.Tensor) -> torch.Tensor:
        """(B,1) -> (B,128): Dense -> swish -> Dense -> Dense (…unet_2d_ising.py:36-38,56)."""
        (w0, b0), (w1, b1), (w2, b2) = self.dense
        e = F.silu(t_norm @ w0 + b0) @ w1 + b1
        return e @ w2 + b2

    def __call__(self, x_nhwc: torch.Tensor, t_norm: torch.Tensor) -> torch.Tensor:
        """x (B,H,W,C), t_norm (B,1) -> score (B,H,W,C)."""
        x = x_nhwc.to(self.dtype).permute(0, 3, 1, 2)
        h, w = x.shape[2], x.shape[3]
        ph, pw = (4 - h % 4) % 4, (4 - w % 4) % 4
        x = F.pad(x, (0, pw, 0, ph))
        ta = self.time_add(t_norm.to(self.dtype))
        x = self._conv(0, x)
        s1 = self._conv(1, x)
        x = F.avg_pool2d(s1, 2)
        x = self._conv(2, x)
        s2 = self._conv(3, x)
        x = F.avg_pool2d(s2, 2)
        x = self._conv(4, x) + ta[:, :, None, None]
        x = self._conv(5, x)
        x = torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), s2], 1)
        x = self._conv(6, x)
        x = self._conv(7, x)
        x = torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), s1], 1)
        x = self._conv(8, x)
        x = self._conv(9, x)
        x = self._conv(10, x)
        return x[:, :, :h, :w].permute(0, 2, 3, 1).contiguous()


def beta_schedule(T: int = 1000) -> np.ndarray:
    """float32(linspace(1e-4, 0.02, T)) (…unet_2d_ising.py:168-172)."""
    return np.linspace(1e-4, 0.02, T).astype(np.float32)


@torch.no_grad()
def diffusion_backward(model: TorchUNet, x_T: np.ndarray, betas: np.ndarray, T: int,
                       noise_z=None, keep_trajectory=False):
    """Reverse sampler with injected standard-normal ``noise_z(t) -> (B,H,W,C)`` arrays.

    noise = z * float32(sqrt(beta_t)); x = (1 + 0.5*beta_t)*x + beta_t*score + noise, left to
    right in float32, noise added at every step including t = 0.
    """
    x = torch.from_numpy(np.asarray(x_T, np.float32)).to(model.dtype)
    b = x.shape[0]
    traj = [x.clone()] if keep_trajectory else None
    for t in range(T - 1, -1, -1):
        t_norm = torch.full((b, 1), np.float32(t) / np.float32(T), dtype=model.dtype)
        score = model(x, t_norm)
        beta = torch.tensor(betas[t], dtype=model.dtype)
        x = (1 + 0.5 * beta) * x + beta * score
        if noise_z is not None:
            z = torch.from_numpy(np.asarray(noise_z(t), np.float32)).to(model.dtype)
            x = x + z * torch.sqrt(beta)
        if keep_trajectory:
            traj.append(x.clone())
    if keep_trajectory:
        return torch.stack(traj[::-1], 0).numpy()
    return x.numpy()


def sign_project(x: np.ndarray) -> np.ndarray:
    """np.where(x > 0, 1, -1); x == 0 -> -1 (…unet_2d_ising.py:211)."""
    return np.where(x > 0, 1, -1).astype(np.int8)


def ising_energy(spins: np.ndarray, J: float, h: float) -> np.ndarray:
    s = spins.astype(np.float64)
    inter = np.sum(s * np.roll(s, -1, 1), (1, 2)) + np.sum(s * np.roll(s, -1, 2), (1, 2))
    return -J * inter - h * np.sum(s, (1, 2))


def ising_magnetization(spins: np.ndarray) -> np.ndarray:
    return np.mean(spins.astype(np.float64), (1, 2))


class TorchBitflipUNet:
    """Independent PyTorch-CPU restatement of ``create_improved_diffusion_unet``
    (``mnist_diffusion/mnist_diffusion.py:162-240``) with library ops (``group_norm``, ``conv2d``,
    ``max_pool2d``): cross-checks ``ref_bitflip.c``, whose arithmetic order is hand-fixed.
    ``weights``: dict of Keras-layout tensors named as at the C boundary."""

    CIN = (32, 32, 64, 192, 96)
    WIDTH = (32, 64, 128, 64, 32)

    def __init__(self, weights: dict, dtype=torch.float64):
        self.dtype = dtype
        self.w = {k: torch.from_numpy(np.asarray(v, np.float64)).to(dtype) for k, v in weights.items()}

    def _conv(self, x, name_w, name_b, pad):
        w = self.w[name_w].permute(3, 2, 0, 1).contiguous()   # (kh,kw,Cin,Cout) -> (Cout,Cin,kh,kw)
        return F.conv2d(x, w, self.w[name_b], padding=pad)

    def _res(self, i, x, temb):
        ci, wd = self.CIN[i], self.WIDTH[i]
        tm = F.silu(temb @ self.w[f"b{i}_t_w"] + self.w[f"b{i}_t_b"])
        h = F.silu(F.group_norm(x, 8, self.w[f"b{i}_g1_g"], self.w[f"b{i}_g1_b"], eps=1e-3))
        h = self._conv(h, f"b{i}_c1_w", f"b{i}_c1_b", 1) + tm[:, :, None, None]
        h = F.silu(F.group_norm(h, 8, self.w[f"b{i}_g2_g"], self.w[f"b{i}_g2_b"], eps=1e-3))
        h = self._conv(h, f"b{i}_c2_w", f"b{i}_c2_b", 1)
        if ci != wd:
            x = self._conv(x, f"b{i}_s_w", f"b{i}_s_b", 0)
        return x + h

    @torch.no_grad()
    def __call__(self, x01: np.ndarray, emb_row: np.ndarray, taps: bool = False):
        """x01 (B,H,W) 0/1; emb_row (32,) sinusoidal embedding of this step -> flip probabilities (B,H,W)."""
        x = torch.from_numpy(np.asarray(x01, np.float64)).to(self.dtype)[:, None]          # NCHW
        e = torch.from_numpy(np.asarray(emb_row, np.float64)).to(self.dtype)[None]
        temb = F.silu(e @ self.w["d1_w"] + self.w["d1_b"])
        temb = F.silu(temb @ self.w["d2_w"] + self.w["d2_b"])
        x = self._conv(x, "in_w", "in_b", 0)
        s1 = self._res(0, x, temb)
        s2 = self._res(1, F.max_pool2d(s1, 2), temb)
        mid = self._res(2, F.max_pool2d(s2, 2), temb)
        d3 = self._res(3, torch.cat([F.interpolate(mid, scale_factor=2, mode="nearest"), s2], 1), temb)
        d4 = self._res(4, torch.cat([F.interpolate(d3, scale_factor=2, mode="nearest"), s1], 1), temb)
        p = torch.sigmoid(self._conv(d4, "out_w", "out_b", 0))[:, 0]
        if taps:
            return p.numpy(), [t.permute(0, 2, 3, 1).numpy() for t in (s1, s2, mid, d3, d4)]
        return p.numpy()
