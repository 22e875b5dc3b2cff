"""Rehearsal of the multi-rank bench on one GPU: two fresh child processes run `bench.py --gpus 2` exactly as the
driver launches it (torch.distributed.run, one rank per process) with the gloo backend, both ranks on device 0.
What must hold for any number of ranks: rank r samples the global lattice indices [r*B, (r+1)*B), so the spins are
bit-identical to one rank sampling [0, 2B); the all-reduced histograms count 2B lattices; rank 0 prints one JSON
line with n_gpus = 2.  (The RCCL leg itself needs two GPUs and runs only on the driver's multi-GPU node.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("L,B", [(32, 96), (64, 24), (128, 6)])
def test_two_rank_bench_matches_one_rank(tmp_path, gpu_real, L, B):
    import torch
    from isingdiff import beta_schedule, sample
    T = 12
    prefix = str(tmp_path / "spins")
    env = dict(os.environ, ISINGDIFF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--batch", str(B), "--L", str(L), "--T", str(T), "--no-cpu-baseline", "--dump-spins", prefix]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["lattices_per_gpu"] == B
    assert rec["observables"]["lattices_in_histogram"] == 2 * B
    assert abs(rec["value"] - 2 * B / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]
    got = np.concatenate([np.load(f"{prefix}.rank{r}.npy") for r in range(2)])
    # one rank, the same global indices (seed of the bench's step 0)
    res = sample(gpu_real, 2 * B, L, T=T, betas=beta_schedule(T), seed=42, precision="bf16", first_global_index=0)
    torch.cuda.synchronize()
    assert np.array_equal(got, res.spins.cpu().numpy())
    assert abs(rec["observables"]["mean_E_per_site"] - res.observables.mean_energy_per_site()) < 1e-12


def _run_bench(tmp_path, tag, ranks, extra):
    prefix = str(tmp_path / tag)
    env = dict(os.environ, ISINGDIFF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())]
    else:
        cmd = [sys.executable]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
            "--dump-spins", prefix] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), np.concatenate([np.load(f"{prefix}.rank{r}.npy") for r in range(ranks)])


@pytest.mark.parametrize("workload,B,extra", [("bitflip", 40, []), ("spectral", 24, ["--T", "12"]), ("dense", 96, ["--T", "12"])])
def test_two_rank_bench_of_the_other_workloads(tmp_path, workload, B, extra):
    """bench.py --workload bitflip | spectral | dense (BASELINE configs 5 and 4, and the dense spectral model) through
    the same multi-rank launch: two ranks of B units equal one rank of 2B units bit for bit (inputs and random numbers
    are keyed by the global index), the all-reduced histogram counts 2B units, one JSON line with `roofline`."""
    two, got2 = _run_bench(tmp_path, "two", 2, ["--workload", workload, "--batch", str(B)] + extra)
    one, got1 = _run_bench(tmp_path, "one", 1, ["--workload", workload, "--batch", str(2 * B)] + extra)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and one["n_gpus"] == 1
    assert two["observables"]["units_in_histogram"] == 2 * B == one["observables"]["units_in_histogram"]
    assert two["roofline"] is not None and two["roofline"]["frac"] > 0
    assert got2.shape == got1.shape and np.array_equal(got2, got1)
    key = "mean_final_pixel" if workload == "bitflip" else "mean_E_per_site"
    assert abs(two["observables"][key] - one["observables"][key]) < 1e-12
    if workload == "bitflip":
        assert two["config"]["T"] == 100 and "full schedule" in two["metric"]
        assert 0.0 < two["observables"][key] < 1.0


def test_in_library_allreduce_single_device():
    """idm_comm_* (the C caller's histogram exchange over RCCL's single-process communicator): with the one device of
    this box the sum is the identity and needs no RCCL; bad device lists are rejected."""
    import ctypes as C
    import torch
    from isingdiff import _lib
    lib = _lib.load()
    h = C.c_void_p()
    devs = (C.c_int * 1)(0)
    _lib.check(lib.idm_comm_create(C.byref(h), devs, 1))
    buf = torch.arange(-5, 3082, dtype=torch.int64, device="cuda")
    want = buf.clone()
    ptrs = (C.c_void_p * 1)(buf.data_ptr())
    _lib.check(lib.idm_allreduce_i64(h, ptrs, buf.numel(), None))
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    _lib.check(lib.idm_comm_destroy(h))
    two = (C.c_int * 2)(0, 0)
    assert lib.idm_comm_create(C.byref(h), two, 2) == -1 and b"twice" in lib.idm_last_error()
    far = (C.c_int * 2)(0, 63)
    assert lib.idm_comm_create(C.byref(h), far, 2) == -1
