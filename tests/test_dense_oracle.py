"""CPU tests of the dense spectral score model (b_diffusion_model_fft_2d_ising.py:19-38,:105-114):
loader, C oracle against a float64 numpy restatement, and the trained-objective known answer.

Parity unpinned by the reference (no tests, no recorded outputs); pinned by the saved weights."""
import os

import numpy as np
import pytest

from isingdiff import dense
from oracle import ref_cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising.keras")
L = 20


@pytest.fixture(scope="module")
def spec():
    return dense.load_dense_spec(PATH)


@pytest.fixture(scope="module")
def ref(spec):
    return ref_cpu.RefDense(spec.kernels, spec.biases, spec.activations)


def np_forward(spec, x, t_norm):
    h = np.concatenate([x.astype(np.float64), np.full((x.shape[0], 1), np.float64(np.float32(t_norm)))], 1)
    for k, b, a in zip(spec.kernels, spec.biases, spec.activations):
        h = h @ k.astype(np.float64) + b.astype(np.float64)
        if a == "tanh":
            h = np.tanh(h)
    return h


def test_loader(spec):
    assert spec.sha256.startswith("597f85f29065ea25")
    assert [k.shape for k in spec.kernels] == [(441, 440)] + [(440, 440)] * 4
    assert spec.activations == ["tanh"] * 4 + ["linear"] and spec.n == 2 * (L // 2 + 1) * L


def test_ctanh():
    L_ = ref_cpu.lib()
    ref_cpu.RefDense([np.zeros((5, 4), np.float32)], [np.zeros(4, np.float32)], ["linear"])   # binds ref_ctanh
    for z in (-30.0, -2.5, -1e-3, 0.0, 1e-4, 0.3, 1.0, 9.0, 100.0):
        assert abs(L_.ref_ctanh(np.float32(z)) - np.tanh(np.float64(np.float32(z)))) < 1.5e-7
    assert L_.ref_ctanh(np.float32(0.0)) == 0.0 and L_.ref_ctanh(np.float32(100.0)) == 1.0


def test_forward_against_float64(spec, ref):
    rs = np.random.RandomState(0)
    x = rs.standard_normal((16, spec.n)).astype(np.float32) * 3
    for t in (0.0, 0.5, 0.999):
        got, want = ref.forward(x, t), np_forward(spec, x, t)
        assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < 2e-6


def spectral_state(spins):
    """ising_fft + [Re | Im] concatenation + flattening (b_diffusion_model_fft_2d_ising.py:93-95, b_fft.py:4-9)."""
    f = np.fft.rfft2(spins.astype(np.float64), axes=(1, 2))
    return np.concatenate([f.real, f.imag], axis=-1).reshape(spins.shape[0], -1).astype(np.float32)


def test_trained_objective_known_answer(spec, ref):
    # target_score = -noise / (1 - alpha_bar_t) (:64).  The saved model is only lightly trained (10 000 steps of
    # batch 64 on unnormalised spectra): measured with a float64 restatement its output explains ~8 % of the
    # target's variance at t >= 900 (corr +0.276, MSE 0.93 against a target variance of 1.00) and almost nothing
    # at t <= 300.  A restatement with wrong layer order, activation or input layout loses that correlation.
    rs = np.random.RandomState(1)
    B = 256
    spins = np.ones((B, L, L))
    spins[rs.rand(B, L, L) < 0.045] = -1
    spins *= rs.choice([-1, 1], size=(B, 1, 1))
    x0 = spectral_state(spins)
    betas = np.linspace(1e-4, 0.02, 1000)
    abar = np.cumprod(1 - betas)
    for t, lo, hi in ((999, 0.22, 0.33), (900, 0.22, 0.33), (600, 0.12, 0.24)):
        eps = rs.standard_normal(x0.shape) * np.sqrt(1 - abar[t])
        xt = (np.sqrt(abar[t]) * x0 + eps).astype(np.float32)
        s = ref.forward(xt, np.float32(t) / np.float32(1000))
        target = -eps / (1 - abar[t])
        c = np.corrcoef(s.ravel(), target.ravel())[0, 1]
        assert lo < c < hi, (t, c)
        if t >= 900:
            assert np.mean((s - target) ** 2) < target.var()


def test_sampler_matches_manual_loop(spec, ref):
    rs = np.random.RandomState(2)
    T, B = 6, 3
    betas = ref_cpu.beta_schedule(1000)[:T]
    x0 = rs.standard_normal((B, spec.n)).astype(np.float32)
    z = rs.standard_normal((T, B, spec.n)).astype(np.float32)
    xf, traj = ref.sample(x0, betas, T, z, trajectory=True)
    x = x0.copy()
    assert np.array_equal(traj[T], x0)
    for t in range(T - 1, -1, -1):
        x = ref_cpu.update(x, ref.forward(x, np.float32(t) / np.float32(T)), z[t], betas[t])
        assert np.array_equal(traj[t], x)
    assert np.array_equal(xf, traj[0])
