"""CPU tests of the MNIST bit-flip path: the C oracle (oracle/ref_bitflip.c) against the independent
PyTorch restatement, and the host-side pieces (tensor table, embedding, synthetic weights).

Parity unpinned: the reference ships no weights or outputs for mnist_diffusion.py; the two
restatements pin each other on synthetic weights."""
import numpy as np
import pytest

from isingdiff import bitflip
from oracle import ref_cpu
from oracle.ref_torch import TorchBitflipUNet


@pytest.fixture(scope="module")
def weights():
    return bitflip.synthetic_weights(0)


@pytest.fixture(scope="module")
def ref(weights):
    return ref_cpu.RefBitflip(weights, bitflip.TENSOR_NAMES)


def test_parameter_count_matches_reference_model(weights):
    # create_improved_diffusion_unet has 569 345 parameters (model.summary() of the reference architecture)
    assert sum(v.size for v in weights.values()) == 569345
    assert len(bitflip.TENSOR_NAMES) == 68
    shapes = bitflip.tensor_shapes()
    assert "b0_s_w" not in shapes and shapes["b3_s_w"] == (1, 1, 192, 64) and shapes["b3_g1_g"] == (192,)


def test_sinusoidal_table():
    emb = bitflip.sinusoidal_table(100)
    assert emb.shape == (100, 32) and emb.dtype == np.float32
    # t = 0: cos = 1, sin = 0; first frequency is 1, last is 1/10000
    assert np.all(emb[0, :16] == 1.0) and np.all(emb[0, 16:] == 0.0)
    t = np.float32(37) / np.float32(100)
    assert abs(emb[37, 0] - np.cos(t)) < 1e-6 and abs(emb[37, 16] - np.sin(t)) < 1e-6
    assert abs(emb[37, 15] - np.cos(t * 1e-4)) < 1e-6 and abs(emb[37, 31] - np.sin(t * 1e-4)) < 1e-6


def test_sinusoidal_table_int64_input_cast():
    """time_input="int64": the declared Input dtype (mnist_diffusion.py:197) applied to float32(t / time_step) in [0, 1)
    truncates every step to position 0, so every row is the embedding of 0."""
    emb = bitflip.sinusoidal_table(100, time_input="int64")
    assert emb.shape == (100, 32)
    assert np.all(emb[:, :16] == 1.0) and np.all(emb[:, 16:] == 0.0)
    assert np.array_equal(emb[0], bitflip.sinusoidal_table(100)[0])
    with pytest.raises(ValueError):
        bitflip.sinusoidal_table(100, time_input="double")


def test_rsqrt_canon_is_accurate():
    for v in (1e-3, 0.0173, 0.5, 1.0, 3.7, 250.0, 1e6):
        r = ref_cpu.rsqrt_canon(v)
        assert abs(r * np.sqrt(np.float64(np.float32(v))) - 1.0) < 3e-7


def test_time_vectors_against_torch(weights, ref):
    emb = bitflip.sinusoidal_table(100)
    tm = TorchBitflipUNet(weights)
    for t in (0, 50, 99):
        got = ref.time_vectors(emb[t])
        e = tm.w["d1_w"].new_tensor(emb[t].astype(np.float64))[None]
        import torch.nn.functional as F
        temb = F.silu(F.silu(e @ tm.w["d1_w"] + tm.w["d1_b"]) @ tm.w["d2_w"] + tm.w["d2_b"])
        want = np.concatenate([F.silu(temb @ tm.w[f"b{i}_t_w"] + tm.w[f"b{i}_t_b"]).numpy()[0] for i in range(5)])
        assert np.max(np.abs(got - want)) < 2e-6


@pytest.mark.parametrize("hw", [(28, 28), (12, 20)])
def test_forward_against_torch(weights, ref, hw):
    H, W = hw
    rs = np.random.RandomState(1)
    x = (rs.rand(3, H, W) < 0.5).astype(np.uint8)
    emb = bitflip.sinusoidal_table(100)
    tm = TorchBitflipUNet(weights)
    for t in (99, 10):
        p, taps = ref.forward(x, emb[t], taps=True)
        pw, tw = tm(x, emb[t], taps=True)
        for a, b in zip(taps, tw):
            assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 2e-5
        assert np.max(np.abs(p - pw)) < 5e-6   # float32 restatement vs float64 library ops
        assert p.min() > 0.0 and p.max() < 1.0


def test_wiring_is_sharp(weights, ref):
    # swapping the concatenation order must change the result (guards the [up, skip] order)
    rs = np.random.RandomState(2)
    x = (rs.rand(2, 28, 28) < 0.5).astype(np.uint8)
    emb = bitflip.sinusoidal_table(100)
    w2 = dict(weights)
    perm = np.concatenate([np.arange(64, 96), np.arange(0, 64)])       # block 4 input: [up 64 | skip 32]
    w2["b4_c1_w"] = weights["b4_c1_w"][:, :, perm, :]
    p = ref.forward(x, emb[5])
    p2 = ref_cpu.RefBitflip(w2, bitflip.TENSOR_NAMES).forward(x, emb[5])
    assert np.max(np.abs(p - p2)) > 1e-3


def test_sampler_xor_semantics(weights, ref):
    rs = np.random.RandomState(3)
    T, B, H, W = 4, 2, 8, 8
    emb = bitflip.sinusoidal_table(T)
    x0 = (rs.rand(B, H, W) < 0.5).astype(np.uint8)
    u = rs.rand(T, B, H, W).astype(np.float32)
    xf, traj = ref.sample(x0, emb, u, trajectory=True)
    assert traj.shape == (T + 1, B, H, W) and np.array_equal(traj[0], x0) and np.array_equal(traj[-1], xf)
    x = x0.copy()
    for k, t in enumerate(range(T - 1, -1, -1)):
        p = ref.forward(x, emb[t])
        x = x ^ (u[t] < p).astype(np.uint8)
        assert np.array_equal(traj[k + 1], x)
