"""GPU parity for the spectral variant (b_diffusion_model_fft_unet_2d_ising.py): BatchNorm-folded
UNet on (Re, Im) spectra, the reverse sampler on a 2-channel state, and irfft2 + sign."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def gpu_fft(fft_spec):
    from isingdiff.model import ScoreModel
    return ScoreModel(fft_spec, device=0, precision="fp32")


@pytest.fixture(scope="module")
def ref_fft(fft_spec):
    from oracle import ref_cpu
    return ref_cpu.RefModel(fft_spec)


def spectra(rng, B, L, t, T=1000):
    x0 = np.ones((B, L, L), np.float32)
    x0[rng.random((B, L, L)) < 0.045] = -1
    x0 *= rng.choice([-1, 1], size=(B, 1, 1)).astype(np.float32)
    s = np.fft.rfft2(x0, s=(L, L))
    xs = np.stack([s.real, s.imag], -1).astype(np.float32)
    ab = np.cumprod(1 - np.linspace(1e-4, 0.02, T))[t]
    eps = rng.standard_normal(xs.shape).astype(np.float32)
    return (np.sqrt(ab) * xs + np.sqrt(1 - ab) * eps).astype(np.float32)


@pytest.mark.parametrize("B,L,t", [(4, 20, 900), (2, 32, 500), (3, 12, 100)])
def test_fft_unet_forward_fp32_bit_exact(gpu_fft, ref_fft, B, L, t):
    """(B, L, L/2+1, 2) state: width 11 -> padded to 12 (:36-44) and cropped; BatchNorm folded."""
    x = spectra(np.random.default_rng(L), B, L, t)
    tn = np.float32(t) / np.float32(1000)
    want, taps = ref_fft.forward(x, tn, taps=True)
    got, gtaps = gpu_fft.forward_taps(x, tn, precision="fp32")
    for i in range(10):
        g = gtaps[i].cpu().numpy()
        bad = np.count_nonzero(g.view(np.uint32) != taps[i].view(np.uint32))
        assert bad == 0, f"layer {i}: {bad} of {g.size} differ"
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_fft_unet_forward_bf16_tolerance(gpu_fft, ref_fft, fft_spec):
    from oracle import ref_cpu
    refq = ref_cpu.RefModel(fft_spec, quant_bf16=True)
    x = spectra(np.random.default_rng(5), 4, 20, 800)
    tn = np.float32(0.8)
    want_q = refq.forward(x, tn)
    want = ref_fft.forward(x, tn)
    got = gpu_fft([x, tn], precision="bf16")
    assert rel_l2(got, want_q) < 1.5e-2
    assert rel_l2(got, want) < 2e-2


@pytest.mark.parametrize("B", [1, 3, 260])
def test_fft_unet_bf16_on_the_transposed_map(gpu_fft, ref_fft, fft_spec, B):
    """At L = 32 the spectrum is 32 x 17 (padded 32 x 20): the bf16 schedule processes it transposed (20 rows x 32
    columns, transposed 3x3 taps, state and score in the caller's layout) so that every MFMA lane is a pixel.  Against
    the oracle within the bf16 tolerances, against the untransposed schedule (option "transpose" = 0) within 5e-3
    (another summation order), also through three sampler steps."""
    from oracle import ref_cpu
    from isingdiff.sampler import beta_schedule, diffusion_backward
    x = spectra(np.random.default_rng(40 + B), B, 32, 700)
    tn = np.float32(0.7)
    got = gpu_fft([x, tn], precision="bf16")
    try:
        gpu_fft.set_option("transpose", 0)
        plain = gpu_fft([x, tn], precision="bf16")
    finally:
        gpu_fft.set_option("transpose", 1)
    assert np.isfinite(got).all() and got.shape == x.shape
    assert rel_l2(got, plain) < 5e-3
    assert not np.array_equal(got, plain)            # (the two schedules do differ: the switch is live)
    if B <= 3:
        refq = ref_cpu.RefModel(fft_spec, quant_bf16=True)
        assert rel_l2(got, refq.forward(x, tn)) < 1.5e-2
        assert rel_l2(got, ref_fft.forward(x, tn)) < 2e-2
        z = np.random.default_rng(1).standard_normal((3,) + x.shape).astype(np.float32)
        betas = beta_schedule(1000)[:3]
        a = diffusion_backward(x, betas, gpu_fft, 3, noise=z, precision="bf16", return_trajectory=False)
        try:
            gpu_fft.set_option("transpose", 0)
            b = diffusion_backward(x, betas, gpu_fft, 3, noise=z, precision="bf16", return_trajectory=False)
        finally:
            gpu_fft.set_option("transpose", 1)
        assert rel_l2(a, b) < 1e-3


def test_first_layer_on_the_matrix_cores_matches_the_vector_alu_kernel(gpu_fft, tmp_path):
    """Layer 0 of the two-channel model (K = 9 taps x 2 channels) runs as six MFMAs per 32 pixels, state and folded weights
    split into bf16 (hi, lo) pairs (hi*hi + hi*lo + lo*hi: 2^-16 relative); ISINGDIFF_CONV0_MFMA=0 selects the kernel that
    does the same sums with float32 vector FMAs.  Both round the activated map to bf16: layer 0's map agrees but for rare roundings, the scores to 5e-3, for the
    plain (L = 20) and the transposed (L = 32) schedule, with batches that fill a 32-pixel tile partly.  (The switch is read
    once per process.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff.keras_file import load_unet_spec\n"
        "from isingdiff.model import ScoreModel\n"
        f"m = ScoreModel(load_unet_spec({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_unet.keras')!r}), device=0, precision='fp32')\n"
        "for tag in ('a', 'b', 'c'):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{tag}.npy'))\n"
        "    np.save(os.path.join(sys.argv[1], f's{tag}.npy'), np.asarray(m([x, np.float32(0.7)], precision='bf16')))\n"
        "    np.save(os.path.join(sys.argv[1], f't{tag}.npy'), m.forward_taps(x, np.float32(0.7), precision='bf16')[1][0].cpu().numpy())\n")
    xs = {"a": spectra(np.random.default_rng(70), 3, 32, 700), "b": spectra(np.random.default_rng(71), 5, 20, 700),
          "c": spectra(np.random.default_rng(72), 130, 32, 700)}
    for tag, x in xs.items():
        np.save(tmp_path / f"x{tag}.npy", x)
    assert "ISINGDIFF_CONV0_MFMA" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_CONV0_MFMA="0"), timeout=240)
    for tag, x in xs.items():
        got = np.asarray(gpu_fft([x, np.float32(0.7)], precision="bf16"))
        want = np.load(tmp_path / f"s{tag}.npy")
        assert np.isfinite(got).all() and got.shape == want.shape
        assert rel_l2(got, want) < 5e-3, tag          # (ten more bf16 layers amplify a flipped rounding, as between any two schedules)
        t0 = gpu_fft.forward_taps(x, np.float32(0.7), precision="bf16")[1][0].cpu().numpy()
        w0 = np.load(tmp_path / f"t{tag}.npy")
        assert np.isfinite(t0).all() and rel_l2(t0, w0) < 2e-4, tag          # layer 0 itself: the same bf16 values but for rare roundings
        assert np.mean(t0 != w0) < 1e-2, tag   # (about one value in 300: sums with cancellation see the 2^-17 of the split)


def test_last_pair_with_the_two_channel_head_fused_matches_the_strip_kernels(gpu_fft, tmp_path):
    """conv8 + conv9 + the two-channel head of the bf16 schedule run as ONE band kernel (8-row bands of the top level, the
    mid tile never leaves LDS); ISINGDIFF_SPECTRAL_BPAIR=0 (read once per process) keeps the two strip launches.  Same
    weights, same operand order, same epilogue expressions: the scores must be identical bit for bit - on the transposed
    map (L = 32: 20 rows, bands of 8 + 8 + 4, the score cropped to 17 rows and written back transposed), on the plain one
    (L = 20: 20 x 12 map) and for a batch that leaves workgroups without work."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff.keras_file import load_unet_spec\n"
        "from isingdiff.model import ScoreModel\n"
        f"m = ScoreModel(load_unet_spec({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_unet.keras')!r}), device=0, precision='fp32')\n"
        "for tag in ('a', 'b', 'c'):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{tag}.npy'))\n"
        "    np.save(os.path.join(sys.argv[1], f's{tag}.npy'), np.asarray(m([x, np.float32(0.3)], precision='bf16')))\n")
    xs = {"a": spectra(np.random.default_rng(80), 3, 32, 300), "b": spectra(np.random.default_rng(81), 5, 20, 300),
          "c": spectra(np.random.default_rng(82), 700, 32, 300)}
    for tag, x in xs.items():
        np.save(tmp_path / f"x{tag}.npy", x)
    assert "ISINGDIFF_SPECTRAL_BPAIR" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_SPECTRAL_BPAIR="0"), timeout=240)
    for tag, x in xs.items():
        got = np.asarray(gpu_fft([x, np.float32(0.3)], precision="bf16"))
        want = np.load(tmp_path / f"s{tag}.npy")
        assert np.isfinite(got).all() and got.shape == want.shape
        assert np.array_equal(got, want), (tag, rel_l2(got, want))


def test_shaped_band_kernels_of_the_transposed_map_are_bit_identical_to_the_run_time_shaped_ones(gpu_fft, tmp_path):
    """At L = 32 the fused pairs of the transposed spectral map (conv2 + conv3 and conv6 + conv7 on 10 x 16, conv8 + conv9 +
    head on 20 x 32 in 8-row bands) run on instances of the band kernel that know map and band size at compile time (+5 %);
    ISINGDIFF_FUSED_SHAPED=0 (read once per process) keeps the run-time-shaped ones.  Same arithmetic, same order: identical
    scores, for a few spectra and for more spectra than workgroups."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff.keras_file import load_unet_spec\n"
        "from isingdiff.model import ScoreModel\n"
        f"m = ScoreModel(load_unet_spec({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_unet.keras')!r}), device=0, precision='fp32')\n"
        "for tag in ('a', 'b'):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{tag}.npy'))\n"
        "    np.save(os.path.join(sys.argv[1], f's{tag}.npy'), np.asarray(m([x, np.float32(0.5)], precision='bf16')))\n")
    xs = {"a": spectra(np.random.default_rng(85), 3, 32, 500), "b": spectra(np.random.default_rng(86), 700, 32, 500)}
    for tag, x in xs.items():
        np.save(tmp_path / f"x{tag}.npy", x)
    assert "ISINGDIFF_FUSED_SHAPED" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_FUSED_SHAPED="0"), timeout=240)
    for tag, x in xs.items():
        got = np.asarray(gpu_fft([x, np.float32(0.5)], precision="bf16"))
        want = np.load(tmp_path / f"s{tag}.npy")
        assert np.isfinite(got).all() and np.array_equal(got, want), (tag, rel_l2(got, want))


def test_fft_unet_sampler_fp32_bit_exact(gpu_fft, ref_fft):
    """Full 1000-step schedule on the 2-channel spectral state with injected noise (this variant's
    dynamics is stable, std(x) stays O(10)), then irfft2 + sign against numpy."""
    import torch
    from isingdiff.sampler import beta_schedule, diffusion_backward
    from isingdiff.spectral import irfft2_sign
    B, L, T = 2, 20, 1000
    rs = np.random.RandomState(3)
    x_T = rs.standard_normal((B, L, L // 2 + 1, 2)).astype(np.float32)
    z = rs.standard_normal((T, B, L, L // 2 + 1, 2)).astype(np.float32)
    betas = beta_schedule(T)
    want = ref_fft.sample(x_T, betas, T, noise_z=z)
    got = diffusion_backward(x_T, betas, gpu_fft, T, noise=z, precision="fp32", return_trajectory=False)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.abs(want).max() < 1e4
    spins = irfft2_sign(torch.from_numpy(got).cuda()).cpu().numpy()
    real = np.fft.irfft2(want[..., 0].astype(np.float64) + 1j * want[..., 1].astype(np.float64), s=(L, L))
    assert np.array_equal(spins, np.where(real > 0, 1, -1))


@pytest.mark.parametrize("L", [4, 10, 20, 32, 64, 128])
def test_irfft2_sign_matches_numpy(L):
    """numpy's axis order: complex ifft along axis 0, then C2R along axis 1 (which drops the imaginary
    parts of the k=0 and k=L/2 columns after the first pass).  Random spectra, non-Hermitian on purpose."""
    import torch
    from isingdiff.spectral import irfft2_sign
    rng = np.random.default_rng(L)
    B = 9
    spec = rng.standard_normal((B, L, L // 2 + 1, 2)).astype(np.float32) * 7
    spins, real = irfft2_sign(torch.from_numpy(spec).cuda(), return_real=True)
    want = np.fft.irfft2(spec[..., 0].astype(np.float64) + 1j * spec[..., 1].astype(np.float64), s=(L, L))
    got = real.cpu().numpy()
    assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())
    assert np.array_equal(spins.cpu().numpy(), np.where(want > 0, 1, -1).astype(np.int8))
    # round trip of a real ±1 lattice is exact after the sign
    s0 = rng.choice(np.array([-1.0, 1.0]), size=(B, L, L))
    f = np.fft.rfft2(s0, s=(L, L))
    back = irfft2_sign(torch.from_numpy(np.stack([f.real, f.imag], -1).astype(np.float32)).cuda()).cpu().numpy()
    assert np.array_equal(back, s0.astype(np.int8))


@pytest.mark.parametrize("L", [4, 20, 32, 64, 128])
def test_irfft2_sign_matches_reference_fixture(L):
    """tests/golden/reference_ifft_L*.npz: outputs of the reference's own ising_ifft (b_fft.py:12-17) on random
    half spectra, written in the build container by tests/golden/make_reference_fixtures.py."""
    import os
    import torch
    from isingdiff.spectral import irfft2_sign
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"reference_ifft_L{L}.npz"))
    want = g["ising_ifft"]
    spins, real = irfft2_sign(torch.from_numpy(g["spectrum"]).cuda(), return_real=True)
    assert np.abs(real.cpu().numpy() - want).max() < 1e-12 * max(1.0, np.abs(want).max())
    assert np.array_equal(spins.cpu().numpy(), np.where(want > 0, 1, -1).astype(np.int8))
    # forward fixture: ising_fft of +-1 lattices (b_fft.py:4-9) comes back as the same lattices
    f = np.stack([g["ising_fft_re"], g["ising_fft_im"]], -1).astype(np.float32)
    assert np.array_equal(irfft2_sign(torch.from_numpy(f).cuda()).cpu().numpy(), g["spins"])


def test_spectral_sample_end_to_end(gpu_fft):
    """sample() on the spectral model: Philox x_T, 1000 bf16 steps, irfft2 + sign, observables."""
    from isingdiff.sampler import sample
    res = sample(gpu_fft, 64, 20, seed=7, precision="bf16", keep_state=True)
    assert res.spins.shape == (64, 20, 20) and res.x_final.shape == (64, 20, 11, 2)
    e = res.observables.mean_energy_per_site()
    m = np.abs(res.observables.magnetization()).mean()
    # the trained spectral model produces ordered lattices (oracle free run: E/L^2 ~ -0.95, <|m|> ~ 0.54)
    assert -2.0 <= e < -0.5, e
    assert m > 0.25, m
