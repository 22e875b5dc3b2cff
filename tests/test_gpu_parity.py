"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle.

Bars: float32 mode is BIT-EXACT against the C oracle (same canonical summation order and
elementary functions); bf16 mode is within a stated relative-L2 tolerance.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BF16_LAYER_TOL = 1.5e-2   # rel-L2 per layer vs the oracle that rounds to bf16 at the same points
BF16_SCORE_TOL = 2e-2    # rel-L2 of the score vs the float32 oracle


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def ising_like(rng, B, H, W, t, T=1000):
    """x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps for mostly-aligned ±1 lattices."""
    x0 = np.ones((B, H, W, 1), np.float32)
    x0[rng.random((B, H, W, 1)) < 0.045] = -1
    x0 *= rng.choice([-1, 1], size=(B, 1, 1, 1)).astype(np.float32)
    ab = np.cumprod(1 - np.linspace(1e-4, 0.02, T))[t]
    eps = rng.standard_normal(x0.shape).astype(np.float32)
    return (np.sqrt(ab) * x0 + np.sqrt(1 - ab) * eps).astype(np.float32)


def test_time_table_bit_exact(gpu_real, ref_real):
    got = gpu_real.time_table(1000)
    want = ref_real.time_table(1000)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("B,H,W,t", [(4, 20, 20, 500), (2, 32, 32, 900), (3, 18, 22, 100), (1, 4, 4, 0), (5, 20, 20, 999)])
def test_forward_fp32_bit_exact(gpu_real, ref_real, B, H, W, t):
    rng = np.random.default_rng(B * 1000 + H)
    x = ising_like(rng, B, H, W, t)
    tn = np.float32(t) / np.float32(1000)
    want, taps = ref_real.forward(x, tn, taps=True)
    got, gtaps = gpu_real.forward_taps(x, tn, precision="fp32")
    for i in range(10):
        g = gtaps[i].cpu().numpy()
        assert g.shape == taps[i].shape
        bad = np.count_nonzero(g.view(np.uint32) != taps[i].view(np.uint32))
        assert bad == 0, f"layer {i}: {bad} of {g.size} elements differ, max abs {np.abs(g - taps[i]).max()}"
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("B,H,W,t", [(4, 20, 20, 500), (2, 32, 32, 900), (2, 18, 22, 700)])
def test_forward_bf16_tolerance(gpu_real, real_spec, ref_real, B, H, W, t):
    from oracle import ref_cpu
    refq = ref_cpu.RefModel(real_spec, quant_bf16=True)
    rng = np.random.default_rng(7)
    x = ising_like(rng, B, H, W, t)
    tn = np.float32(t) / np.float32(1000)
    want_q, taps_q = refq.forward(x, tn, taps=True)
    want = ref_real.forward(x, tn)
    got, gtaps = gpu_real.forward_taps(x, tn, precision="bf16")
    seen = 0
    for i in range(10):
        g = gtaps[i].cpu().numpy()
        if np.isnan(g).all():
            continue  # the fused band kernels keep this layer on chip
        seen += 1
        e = rel_l2(g, taps_q[i])
        assert e < BF16_LAYER_TOL, f"layer {i}: rel-L2 {e}"
    assert seen >= 4
    eq, ef = rel_l2(got.cpu().numpy(), want_q), rel_l2(got.cpu().numpy(), want)
    assert eq < BF16_LAYER_TOL, eq
    assert ef < BF16_SCORE_TOL, ef


@pytest.mark.parametrize("B,H,W,t", [(3, 32, 32, 600), (5, 20, 20, 950), (2, 18, 22, 300), (2, 8, 8, 500)])
def test_bf16_fused_matches_per_layer_kernels(gpu_real, B, H, W, t):
    """The fused band kernels and the one-kernel-per-layer path are two schedules of the same
    bf16 arithmetic; they differ only by accumulation order and the hi/lo split of x."""
    rng = np.random.default_rng(11)
    x = ising_like(rng, B, H, W, t)
    tn = np.float32(t) / np.float32(1000)
    try:
        gpu_real.set_option("fused", 1)
        a, ta = gpu_real.forward_taps(x, tn, precision="bf16")
        gpu_real.set_option("fused", 0)
        b, tb = gpu_real.forward_taps(x, tn, precision="bf16")
    finally:
        gpu_real.set_option("fused", 1)
    for i in (1, 3, 5, 7):
        ga, gb = ta[i].cpu().numpy(), tb[i].cpu().numpy()
        assert not np.isnan(gb).any()
        if np.isnan(ga).all():
            continue
        e = rel_l2(ga, gb)
        assert e < 8e-3, f"layer {i}: fused vs per-layer rel-L2 {e}"
    e = rel_l2(a.cpu().numpy(), b.cpu().numpy())
    assert e < BF16_SCORE_TOL, e


@pytest.mark.parametrize("B,L,t", [(3, 32, 600), (2, 64, 800), (1, 128, 400), (3, 40, 500), (5, 48, 600), (2, 96, 700)])
def test_bf16_strip_kernels_match_fused_and_per_layer(gpu_real, B, L, t):
    """Single-convolution strip kernels (used where a fused band would have to shrink) against the fused
    bands (forced where possible) and the per-layer kernels, layer by layer; also the in-kernel update."""
    import torch
    from isingdiff.sampler import diffusion_backward
    rng = np.random.default_rng(L + 1)
    x = ising_like(rng, B, L, L, t)
    tn = np.float32(t) / np.float32(1000)
    try:
        gpu_real.set_option("strip_below_rows", 1000)        # strips wherever a band is not the whole map
        s, ts = gpu_real.forward_taps(x, tn, precision="bf16")
        z = rng.standard_normal((1, B, L, L, 1)).astype(np.float32)
        xs = diffusion_backward(x, np.array([0.013], np.float32), gpu_real, 1, noise=z, precision="bf16",
                                return_trajectory=False)
        gpu_real.set_option("strip_below_rows", 0)           # fused bands of any height that fits
        f, tf_ = gpu_real.forward_taps(x, tn, precision="bf16")
        gpu_real.set_option("fused", 0)
        p, tp = gpu_real.forward_taps(x, tn, precision="bf16")
        xp = diffusion_backward(x, np.array([0.013], np.float32), gpu_real, 1, noise=z, precision="bf16",
                                return_trajectory=False)
    finally:
        gpu_real.set_option("fused", 1)
        gpu_real.set_option("strip_below_rows", 8)
    for i in (1, 3, 5, 7):
        a, b = ts[i].cpu().numpy(), tp[i].cpu().numpy()
        assert not np.isnan(a).any() and rel_l2(a, b) < 8e-3, f"layer {i}"
    s, f, p = (v.cpu().numpy() for v in (s, f, p))
    assert rel_l2(s, p) < BF16_SCORE_TOL and rel_l2(s, f) < BF16_SCORE_TOL
    xs, xp = (np.asarray(v.cpu() if hasattr(v, "cpu") else v) for v in (xs, xp))
    assert rel_l2(xs, xp) < 1e-3


@pytest.mark.parametrize("L", [64, 128])
def test_large_lattices(gpu_real, real_spec, ref_real, L):
    """The L=20-trained weights applied convolutionally at L=64 / 128 (the large-lattice configuration):
    fp32 bit-exact, bf16 within tolerance."""
    from oracle import ref_cpu
    rng = np.random.default_rng(L)
    x = ising_like(rng, 1, L, L, 700)
    tn = np.float32(0.7)
    want = ref_real.forward(x, tn)
    got = gpu_real([x, tn], precision="fp32")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    want_q = ref_cpu.RefModel(real_spec, quant_bf16=True).forward(x, tn)
    got16 = gpu_real([x, tn], precision="bf16")
    assert rel_l2(got16, want_q) < BF16_LAYER_TOL
    assert rel_l2(got16, want) < BF16_SCORE_TOL


def test_call_and_predict_interfaces(gpu_real, ref_real):
    import torch
    rng = np.random.default_rng(3)
    x = ising_like(rng, 6, 20, 20, 800)
    tn = np.full((6, 1), np.float32(800) / np.float32(1000), np.float32)
    want = ref_real.forward(x, tn[0, 0])
    out_np = gpu_real([x, tn])
    assert isinstance(out_np, np.ndarray)
    assert np.array_equal(out_np, want)
    out_t = gpu_real([torch.from_numpy(x).cuda(), torch.from_numpy(tn).cuda()])
    assert isinstance(out_t, torch.Tensor) and out_t.is_cuda and np.array_equal(out_t.cpu().numpy(), want)
    assert np.array_equal(gpu_real.predict([x, tn], batch_size=4), want)
    with pytest.raises(ValueError):
        gpu_real([x, np.linspace(0, 1, 6, dtype=np.float32)[:, None]])
    with pytest.raises(ValueError):
        gpu_real([x[..., 0], tn])
    assert gpu_real([x[:0], tn[:1]]).shape == (0, 20, 20, 1)


def test_sampler_fp32_injected_noise_bit_exact(gpu_real, ref_real):
    """Deterministic mode, free running for the full schedule: per-step scores, every state of the
    trajectory and the final spins must equal the oracle's bit for bit."""
    from isingdiff.sampler import beta_schedule, diffusion_backward
    from isingdiff.observables import sign_project, observe
    from oracle import ref_cpu
    import torch
    B, L, T = 4, 20, 1000
    rs = np.random.RandomState(0)
    x_T = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    z = rs.standard_normal((T, B, L, L, 1)).astype(np.float32)
    betas = beta_schedule(T)
    want_x, want_scores = ref_real.sample(x_T, betas, T, noise_z=z, record_scores=True)
    traj, scores = diffusion_backward(x_T, betas, gpu_real, T, noise=z, precision="fp32", return_scores=True)
    assert traj.shape == (T + 1, B, L, L, 1)
    assert np.array_equal(traj[T], x_T)
    for t in (999, 900, 500, 100, 0):
        assert np.array_equal(scores[t].view(np.uint32), want_scores[t].view(np.uint32)), f"score at t={t}"
    assert np.array_equal(scores.view(np.uint32), want_scores.view(np.uint32))
    assert np.array_equal(traj[0].view(np.uint32), want_x.view(np.uint32))
    assert np.isfinite(traj[0]).all()
    spins = sign_project(torch.from_numpy(traj[0]).cuda()).cpu().numpy()
    want_spins = ref_cpu.sign_project(want_x)
    assert np.array_equal(spins, want_spins)
    obs = observe(torch.from_numpy(spins[..., 0]).cuda())
    M, S = ref_cpu.observe(want_spins[..., 0])
    assert np.array_equal(obs.M.cpu().numpy(), M) and np.array_equal(obs.S.cpu().numpy(), S)


def test_sampler_ddpm_update_rule(gpu_real, ref_real):
    """The reference's alternative update (b_diffusion_model_unet_2d_ising.py:95, commented out there), selected with
    update_rule="ddpm": float32 free run bit-exact against the oracle's restatement of that line, and - unlike the
    update the script runs (:96), which diverges to ~1e24 - it stays bounded, because it is the ancestral DDPM step
    for the noise-predicting model the saved weights are.  bf16 (fused kernels, Philox noise) gives ordered lattices."""
    from isingdiff.sampler import alpha_cumprod, beta_schedule, diffusion_backward, sample
    B, L, T = 2, 20, 1000
    rs = np.random.RandomState(5)
    x_T = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    z = rs.standard_normal((T, B, L, L, 1)).astype(np.float32)
    betas = beta_schedule(T)
    want = ref_real.sample(x_T, betas, T, noise_z=z, alpha_cumprod=alpha_cumprod(betas))
    got = diffusion_backward(x_T, betas, gpu_real, T, noise=z, precision="fp32", return_trajectory=False, update_rule="ddpm")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.abs(want).max() < 50.0
    other = diffusion_backward(x_T, betas, gpu_real, T, noise=z, precision="fp32", return_trajectory=False)
    assert np.abs(other).max() > 1e10
    res = sample(gpu_real, 512, 32, T=T, seed=3, precision="bf16", update_rule="ddpm", keep_state=True)
    assert float(res.x_final.abs().max()) < 1e3        # (the other rule reaches 1e24; a few outliers of O(50) at L = 32)
    assert res.observables.mean_energy_per_site() < -1.0
    with pytest.raises(ValueError):
        sample(gpu_real, 4, 20, T=10, update_rule="euler")


@pytest.mark.parametrize("B,L", [(4, 20), (2, 32)])
def test_sampler_matches_golden(gpu_real, B, L):
    """Committed fixtures (tests/golden/make_golden.py), RandomState(0) noise, the full 1000-step free run in float32:
    B=4 at L=20 (the size the weights were trained at) and B=2 at L=32 (the headline lattice size).  Scores at four
    steps, the final state, spins, M and S, bit for bit."""
    import os
    import torch
    from isingdiff.sampler import beta_schedule, diffusion_backward
    from isingdiff.observables import observe
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"realspace_B{B}_L{L}_seed0.npz"))
    T = 1000
    rs = np.random.RandomState(0)
    x_T = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    z = rs.standard_normal((T, B, L, L, 1)).astype(np.float32)
    traj, scores = diffusion_backward(x_T, beta_schedule(T), gpu_real, T, noise=z, precision="fp32", return_scores=True)
    for t in (999, 900, 500, 0):
        assert np.array_equal(scores[t].reshape(B, -1), g[f"score_t{t}"]), t
    assert np.array_equal(traj[0].reshape(B, -1).view(np.uint32), g["x_final"].view(np.uint32))
    spins = np.where(traj[0] > 0, 1, -1).astype(np.int8).reshape(B, L, L)
    assert np.array_equal(spins, g["spins"])
    obs = observe(torch.from_numpy(spins).cuda())
    assert np.array_equal(obs.M.cpu().numpy(), g["M"]) and np.array_equal(obs.S.cpu().numpy(), g["S"])


def test_sampler_bf16_teacher_forced(gpu_real, ref_real):
    """bf16 scores along the oracle's own float32 trajectory (teacher forcing): rel-L2 per step."""
    from isingdiff.sampler import beta_schedule
    from oracle import ref_cpu
    B, L, T = 4, 20, 1000
    rs = np.random.RandomState(1)
    x = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    betas = beta_schedule(T)
    worst = 0.0
    for t in range(T - 1, -1, -1):
        tn = np.float32(t) / np.float32(T)
        s_ref = ref_real.forward(x, tn)
        if t % 50 == 0 or t == T - 1:
            s_gpu = gpu_real([x, tn], precision="bf16")
            worst = max(worst, rel_l2(s_gpu, s_ref))
        x = ref_cpu.update(x, s_ref, rs.standard_normal(x.shape).astype(np.float32), betas[t])
    assert worst < BF16_SCORE_TOL, worst


def test_update_bit_exact():
    import ctypes as C
    import torch
    from isingdiff import _lib
    from oracle import ref_cpu
    rng = np.random.default_rng(5)
    for n in (1, 7, 400, 4099):
        x = (rng.standard_normal(n) * 1e3).astype(np.float32)
        s = (rng.standard_normal(n) * 1e2).astype(np.float32)
        z = rng.standard_normal(n).astype(np.float32)
        for beta in [np.float32(1e-4), np.float32(0.02), np.float32(0.0123456)] + list(np.linspace(1e-4, 0.02, 1000).astype(np.float32)[::37]):
            want = ref_cpu.update(x, s, z, beta)
            xd, sd, zd = (torch.from_numpy(a).cuda() for a in (x.copy(), s, z))
            _lib.check(_lib.load().idm_update(0, xd.data_ptr(), sd.data_ptr(), zd.data_ptr(), C.c_float(beta), n, None))
            torch.cuda.synchronize()
            assert np.array_equal(xd.cpu().numpy().view(np.uint32), want.view(np.uint32))
            want0 = ref_cpu.update(x, s, None, beta)
            xd = torch.from_numpy(x.copy()).cuda()
            _lib.check(_lib.load().idm_update(0, xd.data_ptr(), sd.data_ptr(), None, C.c_float(beta), n, None))
            torch.cuda.synchronize()
            assert np.array_equal(xd.cpu().numpy().view(np.uint32), want0.view(np.uint32))


def _philox_normals_cpu(seed, lattice, elems, t):
    from oracle import ref_cpu
    out = np.empty(((elems + 3) // 4) * 4, np.float64)
    for q in range((elems + 3) // 4):
        r = ref_cpu.philox4x32([q, lattice & 0xFFFFFFFF, lattice >> 32, t], [seed & 0xFFFFFFFF, seed >> 32])
        u = ((r >> 9).astype(np.float64) + 0.5) * 2.0 ** -23
        m0, m1 = np.sqrt(-2 * np.log(u[0])), np.sqrt(-2 * np.log(u[2]))
        out[4 * q:4 * q + 4] = [m0 * np.cos(2 * np.pi * u[1]), m0 * np.sin(2 * np.pi * u[1]),
                                m1 * np.cos(2 * np.pi * u[3]), m1 * np.sin(2 * np.pi * u[3])]
    return out[:elems]


def test_philox_stream_matches_counter_definition():
    """Noise depends only on (seed, global lattice, element, t): check against the C oracle's
    Philox4x32-10 + Box-Muller in float64, and check shard independence."""
    import torch
    from isingdiff import _lib
    seed, elems, t = 0x1234567890ABCDEF, 401, 17
    out = torch.empty((5, elems), dtype=torch.float32, device="cuda")
    _lib.check(_lib.load().idm_philox_normal(0, out.data_ptr(), 5, elems, seed, 2 ** 33 + 3, t, None))
    part = torch.empty((2, elems), dtype=torch.float32, device="cuda")
    _lib.check(_lib.load().idm_philox_normal(0, part.data_ptr(), 2, elems, seed, 2 ** 33 + 5, t, None))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[2:4], part.cpu().numpy())
    for b in (0, 4):
        want = _philox_normals_cpu(seed, 2 ** 33 + 3 + b, elems, t)
        assert np.allclose(o[b], want, atol=2e-5, rtol=1e-5)
    big = torch.empty((4096, 400), dtype=torch.float32, device="cuda")
    _lib.check(_lib.load().idm_philox_normal(0, big.data_ptr(), 4096, 400, 42, 0, 3, None))
    v = big.double()
    assert abs(float(v.mean())) < 4e-3 and abs(float(v.var()) - 1) < 6e-3
    assert abs(float((v ** 4).mean()) - 3) < 0.05


@pytest.mark.parametrize("L", [4, 16, 20, 32, 48, 64, 128])
def test_observables_exact(L):
    import torch
    from isingdiff.observables import observe
    from oracle import ref_cpu
    rng = np.random.default_rng(L)
    B = 37 if L > 16 else 5000          # the large batch exercises the persistent loop and the LDS histograms
    spins = rng.choice(np.array([-1, 1], np.int8), size=(B, L, L))
    spins[0] = 1
    spins[1] = -1
    yy, xx = np.meshgrid(np.arange(L), np.arange(L), indexing="ij")
    spins[2] = np.where((yy + xx) % 2 == 0, 1, -1)
    obs = observe(torch.from_numpy(spins).cuda())
    M, S = ref_cpu.observe(spins)
    assert np.array_equal(obs.M.cpu().numpy(), M) and np.array_equal(obs.S.cpu().numpy(), S)
    assert M[0] == L * L and S[0] == 2 * L * L and S[2] == -2 * L * L
    assert obs.energy(1.0, 0.0)[0] == -2 * L * L
    hM = np.bincount((M + L * L) // 2, minlength=L * L + 1)
    hS = np.bincount((S + 2 * L * L) // 2, minlength=2 * L * L + 1)
    assert np.array_equal(obs.hist_M.cpu().numpy(), hM) and np.array_equal(obs.hist_S.cpu().numpy(), hS)
    mom = obs.moments.cpu().numpy()
    assert list(mom) == [B, M.sum(), np.abs(M).sum(), (M ** 2).sum(), S.sum(), (S ** 2).sum()]
    # the reference's numpy formulas
    s = spins.astype(np.float64)
    e_np = -(np.sum(s * np.roll(s, -1, 1), (1, 2)) + np.sum(s * np.roll(s, -1, 2), (1, 2)))
    assert np.array_equal(obs.energy(1.0, 0.0), e_np)
    assert np.allclose(obs.magnetization(), s.mean((1, 2)))


@pytest.mark.parametrize("L", [4, 20, 32, 64, 128])
def test_observables_match_reference_outputs(L):
    """idm_observe against OUTPUTS of the reference's own ising_energy / ising_magnetization
    (b_pure_metoropolis_2d_ising.py:42-60, run on these lattices by tests/golden/make_reference_fixtures.py)."""
    import os
    import torch
    from isingdiff.observables import observe, ising_energy, ising_magnetization
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_observables.npz"))
    spins = g[f"spins_L{L}"]
    obs = observe(torch.from_numpy(spins).cuda())
    assert np.array_equal(obs.energy(1.0, 0.0), g[f"energy_J1_h0_L{L}"])
    assert np.allclose(obs.energy(0.5, 0.3), g[f"energy_J05_h03_L{L}"], rtol=0, atol=1e-9)
    assert np.allclose(obs.magnetization(), g[f"magnetization_L{L}"], rtol=0, atol=1e-15)
    # the drop-in functions with the reference's names and argument order
    assert np.array_equal(ising_energy(spins, 1.0, 0.0), g[f"energy_J1_h0_L{L}"])
    assert np.allclose(ising_magnetization(spins), g[f"magnetization_L{L}"], rtol=0, atol=1e-15)


def test_sample_philox_shard_independent(gpu_real):
    """sample() on [0,6) equals sample() on [0,3) and [3,6) with global indices (fp32, short schedule)."""
    from isingdiff.sampler import sample, beta_schedule
    T = 20
    betas = beta_schedule(1000)[:T]
    full = sample(gpu_real, 6, 20, T=T, betas=betas, seed=9, precision="fp32", keep_state=True)
    a = sample(gpu_real, 3, 20, T=T, betas=betas, seed=9, precision="fp32", keep_state=True)
    b = sample(gpu_real, 3, 20, T=T, betas=betas, seed=9, precision="fp32", keep_state=True, first_global_index=3)
    import torch
    assert torch.equal(full.x_final[:3], a.x_final) and torch.equal(full.x_final[3:], b.x_final)
    assert torch.equal(full.spins[3:], b.spins)
    c = sample(gpu_real, 6, 20, T=T, betas=betas, seed=9, precision="fp32", keep_state=True, chunk=4)
    assert torch.equal(full.x_final, c.x_final)
    assert int(full.observables.moments[0]) == 6


def test_full_size_batch_properties(gpu_real):
    """The bench workload's batch (65 536 lattices, L=32, bf16 fused kernels, automatic chunking) on a short
    slice of the schedule, checked through size-independent properties: a lattice's result depends only on
    (seed, global index) - not on the batch, its chunk or its position in the persistent grid - and the
    histograms and moments equal those recomputed from the spins."""
    import torch
    from isingdiff.sampler import sample, beta_schedule
    n, L, T = 65536, 32, 3
    betas = beta_schedule(1000)[:T]
    full = sample(gpu_real, n, L, T=T, betas=betas, seed=11, precision="bf16", keep_state=True)
    assert full.x_final.shape == (n, L, L, 1) and bool(torch.isfinite(full.x_final).all())
    # windows that straddle every chunk boundary of the plan (taken from the library: the last lattices of one chunk
    # and the first of the next), both ends of the batch and the middle of the first chunk
    chunk, _ = gpu_real.plan_chunk(n, L, L, precision="bf16")
    assert 1024 <= chunk < n and n % chunk == 0, chunk          # the bench batch runs as several equal chunks
    windows = [(0, 5), (chunk // 2 - 3, 7), (n - 5, 5)] + [(k - 4, 7) for k in range(chunk, n, chunk)]
    for g0, cnt in windows:
        part = sample(gpu_real, cnt, L, T=T, betas=betas, seed=11, precision="bf16", keep_state=True, first_global_index=g0)
        assert torch.equal(full.x_final[g0:g0 + cnt], part.x_final), g0
        assert torch.equal(full.spins[g0:g0 + cnt], part.spins), g0
    # batches smaller and larger than the persistent grid (512-768 workgroups), odd sizes
    for cnt in (1, 257, 1000, 3001):
        part = sample(gpu_real, cnt, L, T=T, betas=betas, seed=11, precision="bf16", keep_state=True)
        assert torch.equal(full.x_final[:cnt], part.x_final), cnt
    s = full.spins.to(torch.int32)
    assert bool((s.abs() == 1).all())
    M = s.sum((1, 2))
    S = (s * torch.roll(s, -1, 1)).sum((1, 2)) + (s * torch.roll(s, -1, 2)).sum((1, 2))
    obs = full.observables
    assert torch.equal(obs.M.to(torch.int64), M.to(torch.int64)) and torch.equal(obs.S.to(torch.int64), S.to(torch.int64))
    assert int(obs.hist_M.sum()) == n and int(obs.hist_S.sum()) == n
    hM = torch.bincount(((M + L * L) // 2).to(torch.int64), minlength=L * L + 1)
    assert torch.equal(obs.hist_M.to(torch.int64).cpu(), hM.cpu())
    mom = [int(v) for v in obs.moments.cpu()]
    assert mom == [n, int(M.sum()), int(M.abs().sum()), int((M.to(torch.int64) ** 2).sum()), int(S.sum()),
                   int((S.to(torch.int64) ** 2).sum())]


@pytest.mark.parametrize("L", [32, 64])
def test_bf16_update_is_the_canonical_expression_of_its_score(gpu_real, L):
    """One reverse step on the bf16 path (L=32: fused band kernels, L=64: strip kernels), injected noise: the new
    state equals ((c1*x) + (beta*score)) + (z*sd) with every operation rounded to float32 separately, bit for bit,
    for the score the same call returns (the update is fused into the last convolution kernel)."""
    from isingdiff.sampler import diffusion_backward, beta_schedule
    rs = np.random.RandomState(12)
    B = 6
    x = rs.standard_normal((B, L, L, 1)).astype(np.float32)
    z = rs.standard_normal((1, B, L, L, 1)).astype(np.float32)
    for beta in (beta_schedule(1000)[0], beta_schedule(1000)[999]):
        betas = np.array([beta], np.float32)
        traj, scores = diffusion_backward(x, betas, gpu_real, 1, noise=z, precision="bf16", return_scores=True)
        new, sc = np.asarray(traj[0], np.float32), np.asarray(scores[0], np.float32)
        c1 = np.float32(1.0) + np.float32(0.5) * beta
        sd = np.sqrt(beta, dtype=np.float32)
        want = (c1 * x + beta * sc) + z[0] * sd       # numpy float32: one rounding per operation
        assert want.dtype == np.float32
        assert np.array_equal(new.view(np.uint32), want.view(np.uint32))


# fused_B's switches (read once per process).  The merged-stream kernel (default) gives the same bits as the alternating
# one; the ninth conv8 row of a top band on the copy wavefronts (default) sums that row in another order, so every
# bit-identity comparison with the four-wavefront kernels turns it off.
B_PAIR_ALT = "0"
B_TOP9_ALT = "0"
B_TOP9_OFF = {"ISINGDIFF_B_TOP9": "0"}


def test_copy_wavefront_kernels_are_bit_identical_to_the_four_wavefront_ones(gpu_real, tmp_path):
    """At L=32 the bf16 sampler runs conv6 + conv7 and conv8 + conv9 + head + update in workgroups of eight wavefronts
    (four for the MFMA loops and epilogues, four for tile copies, noise draw and update; csrc/idm_fused_split.hip).
    ISINGDIFF_M3_SPLIT=0 / ISINGDIFF_B_SPLIT=0 select the four-wavefront kernels; by default the eight-wavefront kernels also
    finish a piece's epilogue inside the next MFMA loop (ISINGDIFF_M3_DEFER=0 / ISINGDIFF_B_DEFER=0: right after its own).
    Same partition, arithmetic and Philox counters: the final state must be the same bit for bit.  Odd and small batches
    cover workgroups whose second lattice stream runs dry, 2 049 a workgroup with five lattices in one stream and four in the
    other.  (The switches are read once per process, hence the child processes.)"""
    import os, subprocess, sys
    from isingdiff.sampler import sample, beta_schedule
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model, sample, beta_schedule\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 5, 777, 2049):\n"
        "    r = sample(m, B, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision='bf16', keep_state=True)\n"
        "    np.save(os.path.join(sys.argv[1], f'{sys.argv[2]}{B}.npy'), r.x_final.cpu().numpy())\n")
    switches = ("ISINGDIFF_M3_SPLIT", "ISINGDIFF_B_SPLIT", "ISINGDIFF_M3_DEFER", "ISINGDIFF_B_DEFER", "ISINGDIFF_M3_MFMA16", "ISINGDIFF_M1_MFMA16", "ISINGDIFF_M2_MFMA16", "ISINGDIFF_B_PAIR", "ISINGDIFF_B_TOP9")
    assert not any(k in os.environ for k in switches)
    # (every variant, and the base it is compared with, runs conv2 ... conv7 on the 32x32x16 MFMA and keeps the ninth conv8 row
    # of a top band with the MFMA wavefronts: the defaults sum in another order and are held to tolerances by the tests
    # below.  The base runs fused_B's merged-stream kernel; "p" the alternating-stream one, "d" that one undeferred.)
    variants = (("base", {}), ("m", {"ISINGDIFF_M3_SPLIT": "0"}), ("b", {"ISINGDIFF_B_SPLIT": "0"}),
                ("d", {"ISINGDIFF_M3_DEFER": "0", "ISINGDIFF_B_DEFER": "0", "ISINGDIFF_B_PAIR": "0"}), ("p", {"ISINGDIFF_B_PAIR": B_PAIR_ALT}))
    for tag, env in variants:
        subprocess.run([sys.executable, "-c", code, str(tmp_path), tag], check=True,
                       env=dict(os.environ, ISINGDIFF_M3_MFMA16="0", ISINGDIFF_M1_MFMA16="0", ISINGDIFF_M2_MFMA16="0", **dict(B_TOP9_OFF, **env)), timeout=240)
    variants = variants[1:]
    for B in (1, 5, 777, 2049):
        want = np.load(tmp_path / f"base{B}.npy")
        for tag, _ in variants:
            got = np.load(tmp_path / f"{tag}{B}.npy")
            assert np.isfinite(got).all()
            assert np.array_equal(got, want), (tag, B)


def test_conv6_conv7_on_the_16x16x32_mfma_match_the_32x32x16_kernel(gpu_real, real_spec, ref_real, tmp_path):
    """By default conv6 and conv7 of the L=32 bf16 sampler run on v_mfma_f32_16x16x32_bf16 (other operand lane maps,
    tiles with a 32-byte pad and a shared zero row, weights packed per 16 couts x 32 channels); ISINGDIFF_M3_MFMA16=0
    selects the 32x32x16 stages.  Same products, another summation order: conv7's map within 4e-3 relative L2 of each
    other (bf16 rounding of conv6's output can move by an ulp), the default's score within the bf16 tolerances of the
    oracle, for batches that leave workgroups with zero, one and several lattices; three sampler steps stay within
    1e-2.  (The switch is read once per process.)"""
    import os, subprocess, sys
    from oracle import ref_cpu
    from isingdiff.sampler import sample, beta_schedule
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model, sample, beta_schedule\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 5, 300):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{B}.npy'))\n"
        "    s, taps = m.forward_taps(x, np.float32(0.6), precision='bf16')\n"
        "    np.save(os.path.join(sys.argv[1], f's{B}.npy'), s.cpu().numpy())\n"
        "    np.save(os.path.join(sys.argv[1], f't{B}.npy'), taps[7].cpu().numpy())\n"
        "r = sample(m, 777, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision='bf16', keep_state=True)\n"
        "np.save(os.path.join(sys.argv[1], 'xs.npy'), r.x_final.cpu().numpy())\n")
    rng = np.random.default_rng(61)
    xs = {}
    for B in (1, 5, 300):
        xs[B] = ising_like(rng, B, 32, 32, 600)
        np.save(tmp_path / f"x{B}.npy", xs[B])
    assert "ISINGDIFF_M3_MFMA16" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_M3_MFMA16="0"), timeout=240)
    refq = ref_cpu.RefModel(real_spec, quant_bf16=True)
    for B in (1, 5, 300):
        want_s, want_t = gpu_real.forward_taps(xs[B], np.float32(0.6), precision="bf16")
        got_s, got_t = np.load(tmp_path / f"s{B}.npy"), np.load(tmp_path / f"t{B}.npy")
        assert np.isfinite(got_s).all() and np.isfinite(got_t).all()
        assert rel_l2(got_t, want_t[7].cpu().numpy()) < 4e-3, B
        assert rel_l2(got_s, want_s.cpu().numpy()) < 1e-2, B
        if B <= 5:
            for s_ in (got_s, want_s.cpu().numpy()):
                assert rel_l2(s_, refq.forward(xs[B], np.float32(0.6))) < BF16_LAYER_TOL
                assert rel_l2(s_, ref_real.forward(xs[B], np.float32(0.6))) < BF16_SCORE_TOL
    want = sample(gpu_real, 777, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision="bf16", keep_state=True)
    assert rel_l2(np.load(tmp_path / "xs.npy"), want.x_final.cpu().numpy()) < 1e-2


# the fused_M1 kernel that is NOT the default (the switch is read once per process)
M1_ALT_ENV = {"ISINGDIFF_M1_MFMA16": "0"}


def test_conv2_conv3_on_the_16x16x32_mfma_match_the_32x32x16_kernel(gpu_real, real_spec, ref_real, tmp_path):
    """conv2 and conv3 of the L=32 bf16 sampler have a v_mfma_f32_16x16x32_bf16 kernel as well (whole 16x16 map, four
    wavefronts, staging tile over the mid tile); M1_ALT_ENV selects the one that is not the default.  Same products, another
    summation order: skip2 within 4e-3 relative L2 of each other, the score within 1e-2 and within the bf16 tolerances of
    the oracle, for batches that leave workgroups with zero, one and several lattices; three sampler steps stay within
    1e-2.  (The switch is read once per process.)"""
    import os, subprocess, sys
    from oracle import ref_cpu
    from isingdiff.sampler import sample, beta_schedule
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model, sample, beta_schedule\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 5, 300):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{B}.npy'))\n"
        "    s, taps = m.forward_taps(x, np.float32(0.6), precision='bf16')\n"
        "    np.save(os.path.join(sys.argv[1], f's{B}.npy'), s.cpu().numpy())\n"
        "    np.save(os.path.join(sys.argv[1], f't{B}.npy'), taps[3].cpu().numpy())\n"
        "r = sample(m, 777, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision='bf16', keep_state=True)\n"
        "np.save(os.path.join(sys.argv[1], 'xs.npy'), r.x_final.cpu().numpy())\n")
    rng = np.random.default_rng(62)
    xs = {}
    for B in (1, 5, 300):
        xs[B] = ising_like(rng, B, 32, 32, 600)
        np.save(tmp_path / f"x{B}.npy", xs[B])
    assert "ISINGDIFF_M1_MFMA16" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, **M1_ALT_ENV), timeout=240)
    refq = ref_cpu.RefModel(real_spec, quant_bf16=True)
    for B in (1, 5, 300):
        want_s, want_t = gpu_real.forward_taps(xs[B], np.float32(0.6), precision="bf16")
        got_s, got_t = np.load(tmp_path / f"s{B}.npy"), np.load(tmp_path / f"t{B}.npy")
        assert np.isfinite(got_s).all() and np.isfinite(got_t).all()
        assert rel_l2(got_t, want_t[3].cpu().numpy()) < 4e-3, B
        assert rel_l2(got_s, want_s.cpu().numpy()) < 1e-2, B
        if B <= 5:
            for s_ in (got_s, want_s.cpu().numpy()):
                assert rel_l2(s_, refq.forward(xs[B], np.float32(0.6))) < BF16_LAYER_TOL
                assert rel_l2(s_, ref_real.forward(xs[B], np.float32(0.6))) < BF16_SCORE_TOL
    want = sample(gpu_real, 777, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision="bf16", keep_state=True)
    assert rel_l2(np.load(tmp_path / "xs.npy"), want.x_final.cpu().numpy()) < 1e-2


# the fused_M2 kernel that is NOT the default (the switch is read once per process)
M2_ALT_ENV = {"ISINGDIFF_M2_MFMA16": "0"}


def test_conv4_conv5_on_lattice_pairs_match_the_one_lattice_kernel(gpu_real, real_spec, ref_real, tmp_path):
    """conv4 and conv5 of the L=32 bf16 sampler have a kernel that runs two lattices per pass on v_mfma_f32_16x16x32_bf16
    (8x8 maps, a weight fragment feeds both); M2_ALT_ENV selects the one that is not the default.  Same products, another
    summation order: conv5's map within 4e-3 relative L2 of each other, the score within 1e-2 and within the bf16
    tolerances of the oracle, for odd and even batches (an odd last lattice is paired with itself); three sampler steps
    stay within 1e-2.  (The switch is read once per process.)"""
    import os, subprocess, sys
    from oracle import ref_cpu
    from isingdiff.sampler import sample, beta_schedule
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model, sample, beta_schedule\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 5, 300):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{B}.npy'))\n"
        "    s, taps = m.forward_taps(x, np.float32(0.6), precision='bf16')\n"
        "    np.save(os.path.join(sys.argv[1], f's{B}.npy'), s.cpu().numpy())\n"
        "    np.save(os.path.join(sys.argv[1], f't{B}.npy'), taps[5].cpu().numpy())\n"
        "r = sample(m, 777, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision='bf16', keep_state=True)\n"
        "np.save(os.path.join(sys.argv[1], 'xs.npy'), r.x_final.cpu().numpy())\n")
    rng = np.random.default_rng(63)
    xs = {}
    for B in (1, 5, 300):
        xs[B] = ising_like(rng, B, 32, 32, 600)
        np.save(tmp_path / f"x{B}.npy", xs[B])
    assert "ISINGDIFF_M2_MFMA16" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, **M2_ALT_ENV), timeout=240)
    refq = ref_cpu.RefModel(real_spec, quant_bf16=True)
    for B in (1, 5, 300):
        want_s, want_t = gpu_real.forward_taps(xs[B], np.float32(0.6), precision="bf16")
        got_s, got_t = np.load(tmp_path / f"s{B}.npy"), np.load(tmp_path / f"t{B}.npy")
        assert np.isfinite(got_s).all() and np.isfinite(got_t).all()
        assert rel_l2(got_t, want_t[5].cpu().numpy()) < 4e-3, B
        assert rel_l2(got_s, want_s.cpu().numpy()) < 1e-2, B
        if B <= 5:
            for s_ in (got_s, want_s.cpu().numpy()):
                assert rel_l2(s_, refq.forward(xs[B], np.float32(0.6))) < BF16_LAYER_TOL
                assert rel_l2(s_, ref_real.forward(xs[B], np.float32(0.6))) < BF16_SCORE_TOL
    want = sample(gpu_real, 777, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision="bf16", keep_state=True)
    assert rel_l2(np.load(tmp_path / "xs.npy"), want.x_final.cpu().numpy()) < 1e-2


def test_ninth_row_of_the_top_band_on_the_copy_wavefronts(gpu_real, tmp_path):
    """fused_B's top band has nine conv8 rows for four MFMA wavefronts; ISINGDIFF_B_TOP9 moves the ninth (global row 8) to
    the copy wavefronts as four 16x16x32-MFMA quadrants.  Only what depends on that row may change, and only by summation
    order: conv9 + head spread it over score rows 7 .. 9, every other score row is the same bit for bit, those three agree to
    4e-3 of their norm and, over many lattices, are not all identical (the other path did run).  Batches with one, several and many lattices per
    workgroup.  (The switch is read once per process.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 5, 600):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{B}.npy'))\n"
        "    s = m.predict([x, np.float32(0.6)], precision='bf16')\n"
        "    np.save(os.path.join(sys.argv[1], f's{B}.npy'), np.asarray(s))\n")
    rng = np.random.default_rng(64)
    xs = {}
    for B in (1, 5, 600):
        xs[B] = ising_like(rng, B, 32, 32, 600)
        np.save(tmp_path / f"x{B}.npy", xs[B])
    assert "ISINGDIFF_B_TOP9" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_B_TOP9=B_TOP9_ALT), timeout=240)
    for B in (1, 5, 600):
        want = np.asarray(gpu_real.predict([xs[B], np.float32(0.6)], precision="bf16")).reshape(B, 32, 32)
        got = np.load(tmp_path / f"s{B}.npy").reshape(B, 32, 32)
        assert np.isfinite(got).all()
        rows = np.ones(32, bool)
        rows[7:10] = False
        assert np.array_equal(got[:, rows], want[:, rows]), B
        assert rel_l2(got[:, 7:10], want[:, 7:10]) < 4e-3, B
        # (the float32 sums differ in their last bits only, which survives conv8's rounding to bf16 for about one value in
        # 20 000: a few lattices may well agree everywhere, 600 do not)
        if B == 600:
            assert not np.array_equal(got[:, 7:10], want[:, 7:10])


def test_merged_fused_b_with_the_ninth_row_hand_over_is_bit_identical_to_the_alternating_kernel(gpu_real, tmp_path):
    """The default fused_B merges the two lattice streams and gives the top band's ninth conv8 row to the copy wavefronts.
    ISINGDIFF_B_PAIR=0 runs the alternating-stream kernel with the same ninth-row arithmetic: same bits after three sampler
    steps, for batches whose second stream runs dry (1, 5), with odd lattice counts per workgroup (777) and with more
    lattices in one stream than the other (2 049); and the score of a forward call (no update) agrees bit for bit as well.
    (The switch is read once per process.)"""
    import os, subprocess, sys
    from isingdiff.sampler import sample, beta_schedule
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model, sample, beta_schedule\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 5, 777, 2049):\n"
        "    r = sample(m, B, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision='bf16', keep_state=True)\n"
        "    np.save(os.path.join(sys.argv[1], f'x{B}.npy'), r.x_final.cpu().numpy())\n"
        "x = np.load(os.path.join(sys.argv[1], 'in.npy'))\n"
        "np.save(os.path.join(sys.argv[1], 'score.npy'), np.asarray(m.predict([x, np.float32(0.6)], precision='bf16')))\n")
    x = ising_like(np.random.default_rng(65), 301, 32, 32, 600)
    np.save(tmp_path / "in.npy", x)
    assert "ISINGDIFF_B_PAIR" not in os.environ and "ISINGDIFF_B_TOP9" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_B_PAIR="0"), timeout=240)
    for B in (1, 5, 777, 2049):
        want = sample(gpu_real, B, 32, T=3, betas=beta_schedule(1000)[:3], seed=5, precision="bf16", keep_state=True)
        got = np.load(tmp_path / f"x{B}.npy")
        assert np.isfinite(got).all()
        assert np.array_equal(got, want.x_final.cpu().numpy()), B
    assert np.array_equal(np.load(tmp_path / "score.npy"), np.asarray(gpu_real.predict([x, np.float32(0.6)], precision="bf16")))


def test_conv4_conv5_whole_map_kernel_at_L64_matches_the_strips(gpu_real, real_spec, ref_real, tmp_path):
    """At L = 64 the third level's maps are 16 x 16: conv4 (+ time embedding) and conv5 run as one whole-map kernel on
    v_mfma_f32_16x16x32_bf16 (eight wavefronts, 64 -> 128 -> 128 channels) instead of two strip launches
    (ISINGDIFF_M2_MFMA16=0).  Same products, another summation order: conv5's map within 4e-3 relative L2, the score within
    1e-2 and within the bf16 tolerance of the float32 oracle, for one, several and many lattices per workgroup.  (The switch
    is read once per process.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for B in (1, 3, 300):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{B}.npy'))\n"
        "    s, taps = m.forward_taps(x, np.float32(0.6), precision='bf16')\n"
        "    np.save(os.path.join(sys.argv[1], f's{B}.npy'), s.cpu().numpy())\n"
        "    np.save(os.path.join(sys.argv[1], f't{B}.npy'), taps[5].cpu().numpy())\n")
    rng = np.random.default_rng(66)
    xs = {}
    for B in (1, 3, 300):
        xs[B] = ising_like(rng, B, 64, 64, 600)
        np.save(tmp_path / f"x{B}.npy", xs[B])
    assert "ISINGDIFF_M2_MFMA16" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_M2_MFMA16="0"), timeout=240)
    for B in (1, 3, 300):
        want_s, want_t = gpu_real.forward_taps(xs[B], np.float32(0.6), precision="bf16")
        got_s, got_t = np.load(tmp_path / f"s{B}.npy"), np.load(tmp_path / f"t{B}.npy")
        assert np.isfinite(got_s).all() and np.isfinite(got_t).all()
        assert not np.array_equal(got_t, want_t[5].cpu().numpy())          # (the two paths do differ: the switch is live)
        assert rel_l2(got_t, want_t[5].cpu().numpy()) < 4e-3, B
        assert rel_l2(got_s, want_s.cpu().numpy()) < 1e-2, B
        if B == 1:
            assert rel_l2(want_s.cpu().numpy(), ref_real.forward(xs[B], np.float32(0.6))) < BF16_SCORE_TOL


def test_shaped_first_pairs_at_L64_and_L128_are_bit_identical_to_the_run_time_shaped_kernels(gpu_real, tmp_path):
    """conv0 + conv1 at L = 64 / 128 and conv2 + conv3 at L = 64 run on instances of the fused band kernel that know the map
    and band size at compile time (16 - 17 % faster: the run-time-shaped instances are bound by their index arithmetic);
    ISINGDIFF_FUSED_SHAPED=0 (read once per process) keeps the run-time-shaped ones.  Same arithmetic in the same order: the
    scores and the skip maps must be identical bit for bit, for one lattice and for more lattices than workgroups."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {os.path.join(root, 'diffusion-model_amd')!r})\n"
        "from isingdiff import load_model\n"
        f"m = load_model({os.path.join(root, 'weights', 'b_diffusion_model_2d_ising_realspace_unet.keras')!r}, device=0, precision='bf16')\n"
        "for tag in ('a', 'b', 'c'):\n"
        "    x = np.load(os.path.join(sys.argv[1], f'x{tag}.npy'))\n"
        "    s, taps = m.forward_taps(x, np.float32(0.4), precision='bf16')\n"
        "    np.save(os.path.join(sys.argv[1], f's{tag}.npy'), s.cpu().numpy())\n"
        "    np.save(os.path.join(sys.argv[1], f't1{tag}.npy'), taps[1].cpu().numpy())\n"
        "    np.save(os.path.join(sys.argv[1], f't3{tag}.npy'), taps[3].cpu().numpy())\n")
    rng = np.random.default_rng(90)
    xs = {"a": ising_like(rng, 1, 64, 64, 400), "b": ising_like(rng, 600, 64, 64, 400), "c": ising_like(rng, 3, 128, 128, 400)}
    for tag, x in xs.items():
        np.save(tmp_path / f"x{tag}.npy", x)
    assert "ISINGDIFF_FUSED_SHAPED" not in os.environ
    subprocess.run([sys.executable, "-c", code, str(tmp_path)], check=True, env=dict(os.environ, ISINGDIFF_FUSED_SHAPED="0"), timeout=240)
    for tag, x in xs.items():
        s, taps = gpu_real.forward_taps(x, np.float32(0.4), precision="bf16")
        assert np.isfinite(s.cpu().numpy()).all()
        assert np.array_equal(s.cpu().numpy(), np.load(tmp_path / f"s{tag}.npy")), tag
        assert np.array_equal(taps[1].cpu().numpy(), np.load(tmp_path / f"t1{tag}.npy")), tag
        assert np.array_equal(taps[3].cpu().numpy(), np.load(tmp_path / f"t3{tag}.npy")), tag
