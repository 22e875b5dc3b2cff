"""GPU parity of the dense spectral score model and its sampler (b_diffusion_model_fft_2d_ising.py:19-38,
:105-114, :142-146) through the C-ABI against the C oracle: float32 canonical arithmetic -> bit-exact; the bf16 MFMA path within a
stated tolerance of the float32 oracle (bf16 has 8 significand bits: relative L2 error of the score <= 1e-2)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "weights", "b_diffusion_model_2d_ising.keras")
L = 20


@pytest.fixture(scope="module")
def setup():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from isingdiff import dense
    from oracle import ref_cpu
    model = dense.load_dense_model(PATH, device=0)
    s = model.spec
    return dense, model, ref_cpu.RefDense(s.kernels, s.biases, s.activations), ref_cpu


def test_forward_bit_exact(setup):
    dense, model, ref, _ = setup
    rs = np.random.RandomState(0)
    for B in (1, 7, 64):                      # ragged against the 8-sample workgroup
        x = (rs.standard_normal((B, model.n)) * 5).astype(np.float32)
        for t in (0.0, 0.25, 0.999):
            got = model.score(x, t).cpu().numpy()
            assert np.array_equal(got, ref.forward(x, t))
            xt = np.concatenate([x, np.full((B, 1), np.float32(t), np.float32)], 1)
            assert np.array_equal(model(xt).cpu().numpy(), got)


def test_call_rejects_mixed_t(setup):
    dense, model, _, _ = setup
    xt = np.zeros((2, model.n + 1), np.float32)
    xt[1, -1] = 0.5
    with pytest.raises(ValueError):
        model(xt)


def test_sampler_trajectory_bit_exact(setup):
    dense, model, ref, ref_cpu = setup
    rs = np.random.RandomState(1)
    T, B = 1000, 5
    betas = ref_cpu.beta_schedule(T)
    x0 = rs.standard_normal((B, model.n)).astype(np.float32)
    z = rs.standard_normal((T, B, model.n)).astype(np.float32)
    traj = dense.diffusion_backward(x0, betas, model, T, noise=z).cpu().numpy()
    xf, tr = ref.sample(x0, betas, T, z, trajectory=True)
    assert traj.shape == (T + 1, B, model.n)
    assert np.array_equal(traj, tr) and np.array_equal(traj[0], xf)
    assert np.all(np.isfinite(traj))


def test_philox_noise_is_shard_independent(setup):
    dense, model, _, ref_cpu = setup
    rs = np.random.RandomState(2)
    x0 = rs.standard_normal((12, model.n)).astype(np.float32)
    betas = ref_cpu.beta_schedule(1000)
    whole = dense.diffusion_backward(x0, betas, model, 50, seed=5, return_trajectory=False).cpu().numpy()
    a = dense.diffusion_backward(x0[:5], betas, model, 50, seed=5, return_trajectory=False).cpu().numpy()
    b = dense.diffusion_backward(x0[5:], betas, model, 50, seed=5, first_global_index=5, return_trajectory=False).cpu().numpy()
    assert np.array_equal(whole, np.concatenate([a, b]))


def test_spectral_to_spins_matches_numpy(setup):
    dense, model, _, _ = setup
    rs = np.random.RandomState(3)
    x = rs.standard_normal((9, model.n)).astype(np.float32)
    spins = dense.spectral_to_spins(model._f32(x), L).cpu().numpy()
    r = x.reshape(9, L, -1)
    want = np.where(np.fft.irfft2(r[..., :L // 2 + 1] + 1j * r[..., L // 2 + 1:], s=(L, L)) > 0, 1, -1)   # :142-146
    assert np.array_equal(spins, want.astype(np.int8))


BF16_REL_L2 = 1e-2     # tolerance of the bf16 MFMA path against the float32 oracle (score, relative L2 per call)


@pytest.fixture(scope="module")
def fast(setup):
    dense, model, ref, ref_cpu = setup
    return dense.load_dense_model(PATH, device=0, precision="bf16")


def test_bf16_forward_within_tolerance(setup, fast):
    dense, model, ref, _ = setup
    rs = np.random.RandomState(3)
    for B in (1, 33, 63, 64, 65, 130):        # ragged against the 64-sample workgroup and its 32-sample tiles
        x = (rs.standard_normal((B, model.n)) * 3).astype(np.float32)
        for t in (0.0, 0.25, 0.999):
            want = ref.forward(x, t)
            got = fast.score(x, t).cpu().numpy()
            assert got.shape == want.shape and np.all(np.isfinite(got))
            rel = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert rel <= BF16_REL_L2, (B, t, rel)
    # the t_norm column reaches the score as in float32 (it is applied in float32 in layer 0's epilogue)
    x = (rs.standard_normal((4, model.n)) * 3).astype(np.float32)
    a, b = fast.score(x, 0.25).cpu().numpy(), fast.score(x, 0.75).cpu().numpy()
    wa, wb = ref.forward(x, np.float32(0.25)), ref.forward(x, np.float32(0.75))
    assert np.linalg.norm((b - a) - (wb - wa)) <= 0.05 * np.linalg.norm(wb - wa)


def test_bf16_sampler_follows_float32(setup, fast):
    dense, model, ref, ref_cpu = setup
    rs = np.random.RandomState(4)
    T, B = 1000, 40
    betas = ref_cpu.beta_schedule(T)
    x0 = rs.standard_normal((B, model.n)).astype(np.float32)
    z = rs.standard_normal((T, B, model.n)).astype(np.float32)
    want = dense.diffusion_backward(x0, betas, model, T, noise=z, return_trajectory=False).cpu().numpy()
    got = dense.diffusion_backward(x0, betas, fast, T, noise=z, return_trajectory=False).cpu().numpy()
    assert np.all(np.isfinite(got))
    # same noise, same start: the two precisions end within a few percent of each other after 1000 steps
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 0.05
    # trajectory layout and the first reverse step (one forward in bf16)
    tr = dense.diffusion_backward(x0, betas, fast, 3, noise=z[:3]).cpu().numpy()
    assert tr.shape == (4, B, model.n) and np.array_equal(tr[3], x0)


def test_bf16_philox_matches_float32_noise_and_sharding(setup, fast):
    dense, model, ref, ref_cpu = setup
    rs = np.random.RandomState(5)
    T, B = 4, 70
    betas = ref_cpu.beta_schedule(1000)[:T]
    x0 = rs.standard_normal((B, model.n)).astype(np.float32)
    whole = dense.diffusion_backward(x0, betas, fast, T, seed=9, return_trajectory=False).cpu().numpy()
    a = dense.diffusion_backward(x0[:33], betas, fast, T, seed=9, return_trajectory=False).cpu().numpy()
    b = dense.diffusion_backward(x0[33:], betas, fast, T, seed=9, first_global_index=33, return_trajectory=False).cpu().numpy()
    assert np.array_equal(whole, np.concatenate([a, b]))
    # both precisions key Philox the same way: with in-kernel noise they stay as close as with injected noise
    f32 = dense.diffusion_backward(x0, betas, model, T, seed=9, return_trajectory=False).cpu().numpy()
    assert np.linalg.norm(whole - f32) / np.linalg.norm(f32) <= 1e-2
