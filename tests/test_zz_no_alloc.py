"""idm_forward_noalloc (the entry point that allocates nothing: its activation workspace is a static device
array) against idm_forward, through the C-ABI: bit-identical scores on two calls in a row, the second with a
smaller input, in both precisions; oversize batches are rejected with a clear error."""
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
