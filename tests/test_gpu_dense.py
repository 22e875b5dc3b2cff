"""GPU parity of the dense spectral score model and its sampler (b_diffusion_model_fft_2d_ising.py:19-38,
:105-114, :142-146) through the C-ABI against the C oracle: float32 canonical arithmetic -> bit-exact."""
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
