"""bench_stub — stands in for `scimlsensitivity_jl_amd` when bench.py runs with HIPADJ_BENCH_STUB=1 (tests/test_bench_launch.py).

TEST INFRASTRUCTURE: it exists so that the launch logic of bench.py (self-launch of N ranks, rank / device checks, the dp all-reduce
carriers and their fallback, the one-line output of rank 0) can run on a machine without a GPU, over gloo.  No adjoint is computed:
`du0` is a copy of `u0` and the shard's `dp` is the column sum of its `u0`, so the all-reduced value is checkable.  bench.py labels
such a line "STUB_no_kernel_ran".

HIPADJ_BENCH_STUB_COMM selects how the stand-in of the library's own communicator (hipadj_comm_*) behaves:
    ok (default)   comm_init_rank / comm_selfcheck succeed, adjoint_dev all-reduces dp over torch.distributed (gloo)
    fail           comm_init_rank raises on rank 1                      -> every rank must fall back to the torch carrier
    hang           comm_init_rank never returns on rank 1               -> bench.py's timeout must turn that into the same fall-back
    badcheck       comm_selfcheck raises on rank 0                      -> fall-back after a successful init
"""
import os
import threading

import numpy as np


def shard_range(n_total, rank, world_size):
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def comm_unique_id():
    return bytes(range(128))


class Engine:
    instances = 0

    def __init__(self, model, alg, ntraj, *a, **kw):
        Engine.instances += 1
        self.N, self.n, self.np = int(ntraj), 3, 3
        self._comm = 0
        self._calls = 0
        self._mode = os.environ.get("HIPADJ_BENCH_STUB_COMM", "ok")

    def use_torch_stream(self): pass
    def set_timing(self, level): pass
    def synchronize(self): pass
    def close(self): pass

    def forward_dev(self, u0, p, out=None):
        self._u0 = u0

    def adjoint_dev(self, dLdu, du0, dp):
        import torch.distributed as dist
        du0.copy_(self._u0)
        dp.copy_(self._u0.sum(dim=0))
        if self._comm:                       # "in-stream" all-reduce of the native carrier
            dist.all_reduce(dp, op=dist.ReduceOp.SUM)
        self._calls += 1

    def comm_init_rank(self, unique_id, nranks, rank):
        assert bytes(unique_id) == comm_unique_id()
        if self._mode == "fail" and rank == 1:
            raise RuntimeError("stub: ncclCommInitRank failed")
        if self._mode == "hang" and rank == 1:
            threading.Event().wait()         # never returns
        self._comm, self._rank = int(nranks), int(rank)

    def comm_selfcheck(self):
        if self._mode == "badcheck" and self._rank == 0:
            raise RuntimeError("stub: all-reduce probe mismatch")

    def comm_overlap(self, on=True):
        self._overlap = bool(on)             # the stand-in all-reduces synchronously either way; bench.py's bookkeeping (drain -> synchronize) is what runs

    def comm_destroy(self):
        self._comm = 0

    def comm_count(self):
        return self._comm

    def stats(self):
        return dict(n=3, np=3, forward_ms_last=0.0, adjoint_calls=self._calls, adjoint_main_kernel_ms_total=0.0,
                    adjoint_algorithmic_bytes=0.0, time_segments=1)
