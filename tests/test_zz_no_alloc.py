"""idm_forward_noalloc (the entry point that allocates nothing: its activation workspace is a static device
array) against idm_forward, through the C-ABI: bit-identical scores on two calls in a row, the second with a
smaller input, in both precisions; oversize batches are rejected with a clear error; a first call on a fresh handle
leaves the device's free memory where it was (idm_forward does not); the call can be captured into a HIP graph and
replayed."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _forward(fn, model, x, t_norm, prec):
    import torch
    from isingdiff import _lib
    score = torch.full_like(x, float("nan"))
    B, H, W, _ = x.shape
    _lib.check(fn(model.handle, x.data_ptr(), C.c_float(t_norm), score.data_ptr(), B, H, W, prec,
                  torch.cuda.current_stream(model.device).cuda_stream))
    torch.cuda.synchronize()
    return score.cpu().numpy()


@pytest.mark.parametrize("prec_name,shapes", [
    ("bf16", [(9, 32, 32), (4, 32, 32)]),        # fused band kernels
    ("bf16", [(3, 64, 64), (2, 20, 20)]),        # fused + strip kernels, then a smaller map
    ("fp32", [(5, 20, 20), (2, 18, 22)]),        # canonical float32 kernels; the second map is ragged (padded to 20 x 24)
])
def test_noalloc_forward_equals_forward(gpu_real, prec_name, shapes):
    import torch
    from isingdiff import _lib
    lib = _lib.load()
    prec = _lib.IDM_BF16 if prec_name == "bf16" else _lib.IDM_FP32
    rs = np.random.RandomState(11)
    for i, (B, H, W) in enumerate(shapes):          # two calls in a row, the second smaller
        x = torch.from_numpy(rs.standard_normal((B, H, W, 1)).astype(np.float32)).cuda()
        t_norm = 0.25 + 0.5 * i
        got = _forward(lib.idm_forward_noalloc, gpu_real, x, t_norm, prec)
        want = _forward(lib.idm_forward, gpu_real, x, t_norm, prec)
        assert np.isfinite(want).all()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (prec_name, B, H, W)


def test_noalloc_rejects_oversize_batch(gpu_real):
    import torch
    from isingdiff import _lib
    lib = _lib.load()
    B, L = 4096, 32                                  # 172 KB of workspace per lattice: far beyond 64 MiB
    x = torch.zeros((B, L, L, 1), dtype=torch.float32, device="cuda")
    score = torch.empty_like(x)
    rc = lib.idm_forward_noalloc(gpu_real.handle, x.data_ptr(), C.c_float(0.5), score.data_ptr(), B, L, L, _lib.IDM_BF16,
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == -1
    msg = lib.idm_last_error().decode()
    assert "static scratch" in msg and "at most" in msg, msg
    # the largest batch the message names is accepted
    nmax = int(msg.split("at most ")[1].split()[0])
    assert 300 < nmax < 500, msg
    xs = x[:nmax].contiguous()
    _lib.check(lib.idm_forward_noalloc(gpu_real.handle, xs.data_ptr(), C.c_float(0.5), score.data_ptr(), nmax, L, L,
                                       _lib.IDM_BF16, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.isfinite(score[:nmax]).all()


def _call(fn, model, x, score, t_norm, prec):
    import torch
    from isingdiff import _lib
    B, H, W, _ = x.shape
    _lib.check(fn(model.handle, x.data_ptr(), C.c_float(t_norm), score.data_ptr(), B, H, W, prec,
                  torch.cuda.current_stream(model.device).cuda_stream))


def test_noalloc_first_call_leaves_device_memory_untouched(real_spec):
    """hipMemGetInfo before and after a FIRST call on a fresh handle and a batch size not seen before: the free byte
    count does not move, while idm_forward on the same handle and shape (the positive control) takes its workspace
    from the device.  (A call on another handle and shape first loads the code object, which the runtime does lazily.)"""
    import torch
    from isingdiff import _lib
    from isingdiff.model import ScoreModel
    lib = _lib.load()
    prec = _lib.IDM_BF16
    warm = ScoreModel(real_spec, device=0, precision="bf16")
    xw = torch.zeros((2, 32, 32, 1), dtype=torch.float32, device="cuda")
    sw = torch.empty_like(xw)
    _call(lib.idm_forward_noalloc, warm, xw, sw, 0.5, prec)
    torch.cuda.synchronize()

    fresh = ScoreModel(real_spec, device=0, precision="bf16")
    B, L = 301, 32                                         # 52 MB of workspace, all of it inside the static array
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((B, L, L, 1)).astype(np.float32)).cuda()
    got, want = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    _call(lib.idm_forward_noalloc, fresh, x, got, 0.37, prec)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, f"idm_forward_noalloc changed free device memory by {free0 - free1} bytes"
    _call(lib.idm_forward, fresh, x, want, 0.37, prec)     # positive control: this one allocates the workspace
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    _, per = fresh.plan_chunk(B, L, L, precision="bf16")
    assert free1 - free2 >= B * per // 2, (free1 - free2, B * per)
    assert torch.equal(got, want) and bool(torch.isfinite(want).all())


@pytest.mark.parametrize("prec_name,B,L", [("bf16", 7, 32), ("fp32", 3, 20)])
def test_noalloc_call_captured_into_a_graph(real_spec, prec_name, B, L):
    """The call issued under stream capture (global mode: a hipMalloc anywhere in the process would invalidate the
    capture and fail the call) on a fresh handle, then replayed twice on new inputs: the replays equal idm_forward."""
    import torch
    from isingdiff import _lib
    from isingdiff.model import ScoreModel
    lib = _lib.load()
    prec = _lib.IDM_BF16 if prec_name == "bf16" else _lib.IDM_FP32
    fresh = ScoreModel(real_spec, device=0, precision=prec_name)
    rs = np.random.RandomState(17)
    x = torch.zeros((B, L, L, 1), dtype=torch.float32, device="cuda")
    score = torch.full_like(x, float("nan"))
    want = torch.empty_like(x)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):                          # capture on torch's side stream, error mode "global"
        _call(lib.idm_forward_noalloc, fresh, x, score, 0.61, prec)
    for k in range(2):
        x.copy_(torch.from_numpy(rs.standard_normal((B, L, L, 1)).astype(np.float32)))
        score.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        _call(lib.idm_forward, fresh, x, want, 0.61, prec)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(score).all())
        assert torch.equal(score, want), (prec_name, k)
