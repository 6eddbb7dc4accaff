"""N > 1 path on CPU: two gloo ranks shard an ensemble by shard_range, compute their dL/dp contributions (with the
oracle standing in for the device engine, which needs a GPU) and all-reduce them with the product's
allreduce_dp / gather_du0 — the result must equal the single-process ensemble gradient."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import scimlsensitivity_jl_amd as sa
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    N = 11
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8 / 3])
    ts = np.linspace(0, 1, 11)
    pr = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0, t1=1.0, dt=0.01, save_times=ts, loss="LSQ_SHIFT", loss_shift=2.0)
    lo, hi = sa.shard_range(N, rank, world)
    du0_l, dp_l, _, _ = pr.adjoint_ensemble(u0[lo:hi], p)
    dp = sa.allreduce_dp(dp_l.copy())
    du0 = sa.gather_du0(torch.from_numpy(du0_l), N)
    if rank == 0:
        rdu0, rdp, _, _ = pr.adjoint_ensemble(u0, p)
        q.put((float(np.max(np.abs(dp - rdp) / np.abs(rdp))), float(np.max(np.abs(du0.numpy() - rdu0)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    err_dp, err_du0 = q.get(timeout=180)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert err_dp < 1e-13 and err_du0 == 0.0


class _RecordingEngine:
    """stands in for the device engine (which needs a GPU): records what init_native_allreduce hands to comm_init_rank"""
    def comm_init_rank(self, unique_id, nranks, rank):
        self.args = (bytes(unique_id), nranks, rank)


def _worker_native_id(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import scimlsensitivity_jl_amd as sa
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = sa.init_native_allreduce(_RecordingEngine())
    q.put(eng.args)
    dist.barrier()
    dist.destroy_process_group()


def test_native_allreduce_setup_ships_one_unique_id_to_every_rank():
    """hipadj_comm_*: rank 0 draws the 128-byte RCCL id from the library, torch.distributed (gloo here) only carries it; every
    rank then joins with (id, world, rank).  The communicator itself needs GPUs (GPU suite: single-rank communicator)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_native_id, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = sorted((q.get(timeout=180) for _ in range(2)), key=lambda a: a[2])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert got[0][0] == got[1][0] and len(got[0][0]) == 128 and any(got[0][0])
    assert [g[1:] for g in got] == [(2, 0), (2, 1)]
