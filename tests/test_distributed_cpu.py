"""world_size-2 gloo test of the multi-rank path: shard ranges cover the batch and the observable
histograms / moments all-reduce to the single-rank result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isingdiff.distributed import allreduce_observables, shard_range
    from isingdiff.observables import Observables
    from oracle import ref_cpu
    L, n = 8, 37
    rng = np.random.default_rng(0)
    spins = rng.choice(np.array([-1, 1], np.int8), size=(n, L, L))
    first, count = shard_range(n, rank, world)
    M, S = ref_cpu.observe(spins[first:first + count])
    hM = torch.from_numpy(np.bincount((M + L * L) // 2, minlength=L * L + 1))
    hS = torch.from_numpy(np.bincount((S + 2 * L * L) // 2, minlength=2 * L * L + 1))
    mom = torch.tensor([count, M.sum(), np.abs(M).sum(), (M ** 2).sum(), S.sum(), (S ** 2).sum()], dtype=torch.int64)
    obs = allreduce_observables(Observables(L, torch.from_numpy(M), torch.from_numpy(S), hM, hS, mom))
    if rank == 0:
        np.savez(out_path, hM=obs.hist_M.numpy(), hS=obs.hist_S.numpy(), mom=obs.moments.numpy(),
                 e=obs.mean_energy_per_site())
    dist.barrier()
    dist.destroy_process_group()


def test_histogram_allreduce_world2(tmp_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PYTHONPATH"] = os.pathsep.join([root, os.path.join(root, "diffusion-model_amd"), os.environ.get("PYTHONPATH", "")])
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    from oracle import ref_cpu
    L, n = 8, 37
    spins = np.random.default_rng(0).choice(np.array([-1, 1], np.int8), size=(n, L, L))
    M, S = ref_cpu.observe(spins)
    assert np.array_equal(got["hM"], np.bincount((M + L * L) // 2, minlength=L * L + 1))
    assert np.array_equal(got["hS"], np.bincount((S + 2 * L * L) // 2, minlength=2 * L * L + 1))
    assert list(got["mom"]) == [n, M.sum(), np.abs(M).sum(), (M ** 2).sum(), S.sum(), (S ** 2).sum()]
    assert abs(float(got["e"]) - (-S.sum() / (n * L * L))) < 1e-12
