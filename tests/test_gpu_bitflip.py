"""GPU parity of the MNIST bit-flip sampler (mnist_diffusion/mnist_diffusion.py:43-52, :162-240),
through the C-ABI, against the C oracle: float32 canonical arithmetic, so everything is bit-exact; the bf16
convolution mode within a stated tolerance of the float32 oracle.

Parity unpinned by the reference (no weights or outputs shipped): synthetic weights, oracle =
oracle/ref_bitflip.c, itself cross-checked against the PyTorch restatement in the CPU suite."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from isingdiff import bitflip
    from oracle import ref_cpu
    w = bitflip.synthetic_weights(0)
    return bitflip, bitflip.BitflipModel(w, device=0, time_step=100), ref_cpu.RefBitflip(w, bitflip.TENSOR_NAMES)


def test_time_vectors_bit_exact(setup):
    bitflip, model, ref = setup
    emb = bitflip.sinusoidal_table(100)
    for t in (0, 1, 42, 99):
        assert np.array_equal(model.time_vectors(t), ref.time_vectors(emb[t]))


@pytest.mark.parametrize("hw", [(28, 28), (12, 20)])
def test_forward_bit_exact_with_taps(setup, hw):
    bitflip, model, ref = setup
    H, W = hw
    rs = np.random.RandomState(5)
    x = (rs.rand(5, H, W) < 0.5).astype(np.uint8)
    emb = bitflip.sinusoidal_table(100)
    for t in (99, 30, 0):
        p, taps = model([x[..., None], np.full(5, np.float32(t) / np.float32(100), np.float32)], taps=True)
        pr, tr = ref.forward(x, emb[t], taps=True)
        for i, (a, b) in enumerate(zip(taps, tr)):
            assert np.array_equal(a.cpu().numpy(), b), f"ResBlock {i} at t={t}"
        assert np.array_equal(p.cpu().numpy()[..., 0], pr)


def test_model_rejects_mixed_or_off_grid_t(setup):
    bitflip, model, _ = setup
    x = np.zeros((2, 28, 28, 1), np.uint8)
    with pytest.raises(ValueError):
        model([x, np.array([0.1, 0.2], np.float32)])
    with pytest.raises(ValueError):
        model([x, np.array([0.123, 0.123], np.float32)])


def test_sampler_trajectory_bit_exact(setup):
    bitflip, model, ref = setup
    rs = np.random.RandomState(6)
    T, B, H, W = 100, 3, 28, 28
    x0 = rs.randint(2, size=(B, H, W)).astype(np.uint8)           # X_asymp = randint(2) (:518)
    u = rs.rand(T, B, H, W).astype(np.float32)
    traj = bitflip.diffusion_backward(x0[..., None], model, T, uniforms=u).cpu().numpy()[..., 0]
    xf, tr = ref.sample(x0, bitflip.sinusoidal_table(T), u, trajectory=True)
    assert traj.shape == (T + 1, B, H, W)
    assert np.array_equal(traj, tr)
    assert np.array_equal(traj[0], x0) and np.array_equal(traj[-1], xf)


def test_sampler_int64_time_input_bit_exact(setup):
    """The other reading of the reference's time input (Input dtype int64, mnist_diffusion.py:197: every step embeds
    position 0): same kernels, a different host-side embedding table; bit-exact against the oracle fed that table, and
    different from the float reading."""
    bitflip, model, ref = setup
    rs = np.random.RandomState(16)
    T, B, H, W = 100, 2, 28, 28
    x0 = rs.randint(2, size=(B, H, W)).astype(np.uint8)
    u = rs.rand(T, B, H, W).astype(np.float32)
    model_i = bitflip.BitflipModel(model.weights, device=0, time_step=T, time_input="int64")
    got = bitflip.diffusion_backward(x0[..., None], model_i, T, uniforms=u, return_trajectory=False).cpu().numpy()[..., 0]
    xf, _ = ref.sample(x0, bitflip.sinusoidal_table(T, time_input="int64"), u, trajectory=True)
    assert np.array_equal(got, xf)
    other = bitflip.diffusion_backward(x0[..., None], model, T, uniforms=u, return_trajectory=False).cpu().numpy()[..., 0]
    assert not np.array_equal(got, other)


def test_sampler_philox_uniforms_do_not_depend_on_sharding_or_chunking(setup):
    bitflip, model, _ = setup
    rs = np.random.RandomState(7)
    x0 = rs.randint(2, size=(6, 28, 28, 1)).astype(np.uint8)
    whole = bitflip.diffusion_backward(x0, model, 100, seed=9, return_trajectory=False).cpu().numpy()
    a = bitflip.diffusion_backward(x0[:4], model, 100, seed=9, return_trajectory=False, chunk=3).cpu().numpy()
    b = bitflip.diffusion_backward(x0[4:], model, 100, seed=9, first_global_index=4, return_trajectory=False).cpu().numpy()
    assert np.array_equal(whole, np.concatenate([a, b]))
    other = bitflip.diffusion_backward(x0, model, 100, seed=10, return_trajectory=False).cpu().numpy()
    assert not np.array_equal(whole, other)


def test_flip_rate_follows_probability(setup):
    # one step from an all-zero image: the fraction of flipped pixels equals the mean probability
    bitflip, model, _ = setup
    import torch
    B = 64
    x0 = np.zeros((B, 28, 28, 1), np.uint8)
    p = model([x0, np.full(B, np.float32(99) / np.float32(100), np.float32)]).cpu().numpy()
    # run the whole schedule is not needed: compare first step through the trajectory
    traj = bitflip.diffusion_backward(x0, model, 100, seed=3).cpu().numpy()
    flipped = traj[1].astype(np.float64).mean()
    assert abs(flipped - p.mean()) < 4 * np.sqrt(p.mean() * (1 - p.mean()) / p.size) + 1e-3


BF16_PROBA_ABS = 2e-2      # bf16 convolution mode: flip probabilities against the float32 oracle, max abs error
BF16_TAP_REL_L2 = 2e-2     # ... and every ResBlock output, relative L2


@pytest.fixture(scope="module")
def fast(setup):
    bitflip, model, ref = setup
    return bitflip.BitflipModel(model.weights, device=0, time_step=100, precision="bf16")


@pytest.mark.parametrize("hw", [(28, 28), (12, 20), (8, 40), (32, 32), (4, 16), (64, 24)])
def test_bf16_forward_within_tolerance(setup, fast, hw):
    """(28, 28) and (12, 20): LDS-tiled convolutions, whole maps and bands, half activations with the GroupNorms folded into
    the staging; (8, 40): wider than a 32-pixel tile row, the per-layer convolution kernel on float32 activations with
    one-workgroup GroupNorm + swish kernels.  (32, 32): the widest tile row; (4, 16): one-row maps at the third level;
    (64, 24): several bands per map at every level.  MFMA shortcuts in all of them."""
    bitflip, model, ref = setup
    H, W = hw
    rs = np.random.RandomState(6)
    x = (rs.rand(5, H, W) < 0.5).astype(np.uint8)
    emb = bitflip.sinusoidal_table(100)
    for t in (99, 30, 0):
        p, taps = fast([x[..., None], np.full(5, np.float32(t) / np.float32(100), np.float32)], taps=True)
        pr, tr = ref.forward(x, emb[t], taps=True)
        for i, (a, b) in enumerate(zip(taps, tr)):
            a = a.cpu().numpy()
            assert np.all(np.isfinite(a))
            assert np.linalg.norm(a - b) <= BF16_TAP_REL_L2 * np.linalg.norm(b), f"ResBlock {i} at t={t}"
        assert np.max(np.abs(p.cpu().numpy()[..., 0] - pr)) <= BF16_PROBA_ABS


def test_bf16_sampler_flips_like_float32(setup, fast):
    bitflip, model, ref = setup
    rs = np.random.RandomState(7)
    B, T = 16, 100
    x0 = (rs.rand(B, 28, 28, 1) < 0.5).astype(np.uint8)
    u = rs.rand(T, B, 28, 28).astype(np.float32)
    a = bitflip.diffusion_backward(x0, model, T, uniforms=u).cpu().numpy()
    b = bitflip.diffusion_backward(x0, fast, T, uniforms=u).cpu().numpy()
    assert a.shape == b.shape == (T + 1, B, 28, 28, 1) and np.array_equal(b[0], x0)
    # same uniforms: a pixel flips differently only where u lies within the probability error of p, so the
    # one-step disagreement stays at the percent level (the states then evolve apart, as any perturbation would)
    assert np.mean(a[1] != b[1]) <= 0.03
    # and the flip statistics agree over the whole schedule
    fa, fb = np.mean(a[1:] != a[:-1]), np.mean(b[1:] != b[:-1])
    assert abs(fa - fb) <= 0.02 * max(fa, 1e-3) + 2e-3
    # Philox uniforms: independent of sharding in bf16 mode too
    w = bitflip.diffusion_backward(x0, fast, T, seed=3, return_trajectory=False).cpu().numpy()
    h1 = bitflip.diffusion_backward(x0[:5], fast, T, seed=3, return_trajectory=False).cpu().numpy()
    h2 = bitflip.diffusion_backward(x0[5:], fast, T, seed=3, first_global_index=5, return_trajectory=False).cpu().numpy()
    assert np.array_equal(w, np.concatenate([h1, h2]))


def test_bf16_activation_storage_matches_float32_storage(setup, fast, tmp_path):
    """The bf16 mode keeps its activations in IEEE half between kernels and folds both GroupNorms of a ResBlock into the
    convolutions' tile staging wherever those kernels cover the shape ((28, 28) and (12, 20) here; (8, 40) is wider than a
    tile row and runs on float32 activations either way).  ISINGDIFF_BITFLIP_ACT=f32 (read once per process, so a child
    runs it) keeps float32 activations and the separate GroupNorm kernels.  Each must meet the mode's tolerance against the
    float32 oracle with room to spare (half storage adds 1/64 of the bf16 operand rounding, not a second one), and the two
    differ from one another by no more than two independent bf16 roundings of the same values can."""
    import os, subprocess, sys
    bitflip, model, ref = setup
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, 'diffusion-model_amd')\n"
        "from isingdiff import bitflip\n"
        "m = bitflip.BitflipModel(bitflip.synthetic_weights(0), device=0, time_step=100, precision='bf16')\n"
        "out = {}\n"
        "for H, W in ((28, 28), (12, 20)):\n"
        "    x = (np.random.RandomState(11).rand(6, H, W, 1) < 0.5).astype(np.uint8)\n"
        "    for t in (99, 0):\n"
        "        p, taps = m([x, np.full(6, np.float32(t) / np.float32(100), np.float32)], taps=True)\n"
        "        out[f'p_{H}_{t}'] = p.cpu().numpy()\n"
        "        for i, a in enumerate(taps): out[f'tap{i}_{H}_{t}'] = a.cpu().numpy()\n"
        "np.savez(sys.argv[1] + '/f32act.npz', **out)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, cwd=root,
                   env=dict(os.environ, ISINGDIFF_BITFLIP_ACT="f32"), timeout=240)
    other = np.load(tmp_path / "f32act.npz")
    emb = bitflip.sinusoidal_table(100)
    differs = False
    for H, W in ((28, 28), (12, 20)):
        x = (np.random.RandomState(11).rand(6, H, W, 1) < 0.5).astype(np.uint8)
        for t in (99, 0):
            p, taps = fast([x, np.full(6, np.float32(t) / np.float32(100), np.float32)], taps=True)
            pr, tr = ref.forward(x[..., 0], emb[t], taps=True)
            p = p.cpu().numpy()
            assert np.max(np.abs(p - other[f"p_{H}_{t}"])) <= 1.5e-2
            assert np.sqrt(np.mean((p - other[f"p_{H}_{t}"]) ** 2)) <= 4e-3
            for q in (p, other[f"p_{H}_{t}"]):
                assert np.max(np.abs(q[..., 0] - pr)) <= 0.75 * BF16_PROBA_ABS
            for i, a in enumerate(taps):
                a, o = a.cpu().numpy(), other[f"tap{i}_{H}_{t}"]
                assert np.linalg.norm(a - o) <= 1e-2 * np.linalg.norm(o), f"ResBlock {i}, {H} x {W}, t={t}"
                for q in (a, o):
                    assert np.linalg.norm(q - tr[i]) <= 0.5 * BF16_TAP_REL_L2 * np.linalg.norm(tr[i])
                differs |= not np.array_equal(a, o)
    assert differs      # (the switch did select another set of kernels)

