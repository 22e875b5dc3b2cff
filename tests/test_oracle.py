"""CPU tests of the oracle: known answers, golden fixture, agreement between the two restatements."""
import os

import numpy as np
import pytest


def _ising_xt(rng, B, L, t, T=1000):
    x0 = np.ones((B, L, L, 1), np.float32)
    x0[rng.random((B, L, L, 1)) < 0.045] = -1
    x0 *= rng.choice([-1, 1], size=(B, 1, 1, 1)).astype(np.float32)
    ab = np.cumprod(1 - np.linspace(1e-4, 0.02, T))[t]
    eps = rng.standard_normal(x0.shape).astype(np.float32)
    return x0, eps, (np.sqrt(ab) * x0 + np.sqrt(1 - ab) * eps).astype(np.float32)


def test_trained_objective_known_answer(ref_real):
    """The real-space UNet was trained to predict +noise (b_diffusion_model_unet_2d_ising.py:124):
    a correct restatement of the layer wiring reproduces corr(model(x_t, t/T), eps) ~ 0.985 at
    t=600 and ~0.999 at t=900 on near-ordered lattices; a wrong concat order gives ~ -0.1."""
    rng = np.random.default_rng(0)
    for t, lo in ((900, 0.995), (600, 0.97)):
        _, eps, xt = _ising_xt(rng, 64, 20, t)
        pred = ref_real.forward(xt, np.float32(t) / np.float32(1000))
        c = np.corrcoef(pred.ravel(), eps.ravel())[0, 1]
        assert c > lo, (t, c)


def test_fft_unet_known_answer(fft_spec):
    """FFT-UNet target is -noise (b_diffusion_model_fft_unet_2d_ising.py:134): corr(model, -eps) > 0.9 at t=900."""
    from oracle import ref_cpu
    ref = ref_cpu.RefModel(fft_spec)
    rng = np.random.default_rng(1)
    x0, _, _ = _ising_xt(rng, 64, 20, 900)
    spec = np.fft.rfft2(x0[..., 0], s=(20, 20))
    xs = np.stack([spec.real, spec.imag], -1).astype(np.float32)
    ab = np.cumprod(1 - np.linspace(1e-4, 0.02, 1000))[900]
    eps = rng.standard_normal(xs.shape).astype(np.float32)
    xt = (np.sqrt(ab) * xs + np.sqrt(1 - ab) * eps).astype(np.float32)
    pred = ref.forward(xt, np.float32(0.9))
    assert pred.shape == xt.shape
    assert np.corrcoef(pred.ravel(), -eps.ravel())[0, 1] > 0.9


@pytest.mark.parametrize("B,H,W", [(3, 20, 20), (2, 32, 32), (2, 18, 22), (1, 4, 4)])
def test_c_oracle_agrees_with_torch_restatement(real_spec, ref_real, B, H, W):
    """Two independent restatements (C canonical order vs library convolutions, float64)."""
    import torch
    from oracle.ref_torch import TorchUNet
    m64 = TorchUNet(real_spec, torch.float64)
    rng = np.random.default_rng(B + H)
    x = rng.standard_normal((B, H, W, 1)).astype(np.float32)
    tn = np.float32(0.37)
    got = ref_real.forward(x, tn)
    with torch.no_grad():
        want = m64(torch.from_numpy(x), torch.full((B, 1), float(tn), dtype=torch.float64)).numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-5


def test_fft_unet_c_vs_torch(fft_spec):
    import torch
    from oracle import ref_cpu
    from oracle.ref_torch import TorchUNet
    ref, m64 = ref_cpu.RefModel(fft_spec), TorchUNet(fft_spec, torch.float64)
    x = np.random.default_rng(2).standard_normal((3, 20, 11, 2)).astype(np.float32)
    got = ref.forward(x, np.float32(0.5))
    with torch.no_grad():
        want = m64(torch.from_numpy(x), torch.full((3, 1), 0.5, dtype=torch.float64)).numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-5


@pytest.mark.parametrize("B,L", [(4, 20), (2, 32)])
def test_oracle_reproduces_golden_fixture(ref_real, B, L):
    """tests/golden/realspace_B4_L20_seed0.npz and realspace_B2_L32_seed0.npz (made by tests/golden/make_golden.py):
    the oracle is deterministic bit for bit, independent of thread count and build."""
    from oracle import ref_cpu
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"realspace_B{B}_L{L}_seed0.npz"))
    T = 1000
    rs = np.random.RandomState(0)
    x_T = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    z = rs.standard_normal((T, B, L, L, 1)).astype(np.float32)
    x, scores = ref_real.sample(x_T, ref_cpu.beta_schedule(T), T, noise_z=z, record_scores=True)
    for t in (999, 900, 500, 0):
        assert np.array_equal(scores[t].reshape(B, -1), g[f"score_t{t}"])
    assert np.array_equal(x.reshape(B, -1), g["x_final"])
    spins = ref_cpu.sign_project(x)[..., 0]
    assert np.array_equal(spins, g["spins"])
    M, S = ref_cpu.observe(spins)
    assert np.array_equal(M, g["M"]) and np.array_equal(S, g["S"])
    # the free-running real-space sampler diverges (|x| ~ 1e24) but stays finite in float32
    assert np.isfinite(x).all() and 1e22 < np.abs(x).max() < 1e26
    assert np.array_equal(ref_real.time_add(np.float32(500) / np.float32(1000)), g["time_add_t500"])


def test_canonical_elementary_functions():
    from oracle import ref_cpu
    lib = ref_cpu.lib()
    xs = np.linspace(-87, 87, 4001).astype(np.float32)
    e = np.array([lib.ref_cexp(float(v)) for v in xs], np.float64)
    assert np.max(np.abs(e - np.exp(xs.astype(np.float64))) / np.exp(xs.astype(np.float64))) < 2e-7
    ys = np.array([-1e24, -100, -87, -20, -1, -1e-3, 0, 1e-3, 1, 20, 100, 1e24], np.float32)
    s = np.array([lib.ref_silu(float(v)) for v in ys], np.float64)
    want = ys.astype(np.float64) / (1 + np.exp(-np.clip(ys.astype(np.float64), -700, 700)))
    assert np.allclose(s, want, rtol=3e-7, atol=1e-13)


def test_update_rule_and_projection():
    from oracle import ref_cpu
    x = np.array([1.5, -2.0, 0.0, 3.0], np.float32)
    s = np.array([0.5, 0.25, -1.0, 2.0], np.float32)
    z = np.array([1.0, -1.0, 0.5, 0.0], np.float32)
    beta = np.float32(0.02)
    want = (np.float32(1) + np.float32(0.5) * beta) * x + beta * s + z * np.sqrt(beta)
    assert np.allclose(ref_cpu.update(x, s, z, beta), want, rtol=1e-6)
    assert np.array_equal(ref_cpu.sign_project(np.array([0.0, -0.0, 1e-30, -1e-30, 5.0], np.float32)),
                          np.array([-1, -1, 1, -1, 1], np.int8))


@pytest.mark.parametrize("L", [4, 20, 32])
def test_observables_match_reference_formulas(L):
    """ising_energy / ising_magnetization as written in b_pure_metoropolis_2d_ising.py:42-60 (np.roll)."""
    from oracle import ref_cpu
    rng = np.random.default_rng(L)
    s = rng.choice(np.array([-1, 1], np.int8), size=(9, L, L))
    s[0] = 1
    M, S = ref_cpu.observe(s)
    f = s.astype(np.float64)
    inter = np.sum(f * np.roll(f, -1, 1), (1, 2)) + np.sum(f * np.roll(f, -1, 2), (1, 2))
    assert np.array_equal(-1.0 * S - 0.0 * M, -1.0 * inter - 0.0 * f.sum((1, 2)))
    assert np.allclose(M / L ** 2, f.mean((1, 2)))
    assert S[0] == 2 * L * L and M[0] == L * L


@pytest.mark.parametrize("L", [4, 20, 32, 64, 128])
def test_observables_match_reference_outputs(L):
    """The oracle's integer sums against OUTPUTS of the reference's own ising_energy / ising_magnetization
    (b_pure_metoropolis_2d_ising.py:42-60; tests/golden/make_reference_fixtures.py took the two definitions out of the
    file's syntax tree and ran them on these lattices): E = -J*S - h*M and m = M / L^2, exactly."""
    import os
    from oracle import ref_cpu
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_observables.npz"))
    spins = g[f"spins_L{L}"]
    M, S = ref_cpu.observe(spins)
    assert np.array_equal(-1.0 * S - 0.0 * M, g[f"energy_J1_h0_L{L}"])
    assert np.array_equal(-1.0 * S, g[f"energy_int8_J1_h0_L{L}"])
    assert np.allclose(-0.5 * S - 0.3 * M, g[f"energy_J05_h03_L{L}"], rtol=0, atol=1e-9)
    assert np.allclose(M / float(L * L), g[f"magnetization_L{L}"], rtol=0, atol=1e-15)
    assert g[f"energy_J1_h0_L{L}"][0] == -2 * L * L and g[f"energy_J1_h0_L{L}"][2] == 2 * L * L


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    from oracle import ref_cpu
    assert list(ref_cpu.philox4x32([0, 0, 0, 0], [0, 0])) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(ref_cpu.philox4x32([0xffffffff] * 4, [0xffffffff] * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(ref_cpu.philox4x32([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_ddpm_update_rule_is_the_stable_one(ref_real):
    """The commented-out update of b_diffusion_model_unet_2d_ising.py:95, restated in ref_update_ddpm: against a float64
    evaluation of the line as written, and as a free run - it is the ancestral DDPM step for a model that predicts the
    noise (which the saved weights do, see the correlation probes above), so the state stays O(1) and orders, where the
    update the script runs (:96) diverges."""
    from oracle import ref_cpu
    from oracle.ref_cpu import beta_schedule
    T = 1000
    noise_beta = np.linspace(1e-4, 0.02, T)
    acp = np.cumprod(1 - noise_beta).astype(np.float32)
    betas = beta_schedule(T)
    rs = np.random.RandomState(5)
    # one step against the formula of :95 in float64
    x = rs.standard_normal(64).astype(np.float32); s = rs.standard_normal(64).astype(np.float32)
    z = rs.standard_normal(64).astype(np.float32)
    t = 700
    got = x.copy()
    ref_cpu.lib().ref_update_ddpm(got.ctypes.data, s.ctypes.data, z.ctypes.data, ref_cpu.C.c_float(betas[t]), ref_cpu.C.c_float(acp[t]), 64)
    b, a = np.float64(betas[t]), np.float64(acp[t])
    want = (1 / np.sqrt(1 - b)) * (x.astype(np.float64) - (b / np.sqrt(1 - a)) * s) + z * np.sqrt(b)
    assert np.abs(got - want).max() < 1e-6
    # free run
    x_T = rs.standard_normal((1, 20, 20, 1)).astype(np.float32)
    zz = rs.standard_normal((T, 1, 20, 20, 1)).astype(np.float32)
    out = ref_real.sample(x_T, betas, T, noise_z=zz, alpha_cumprod=acp)
    assert np.abs(out).max() < 10.0
    assert abs(np.sign(out).mean()) > 0.8
