"""Ensemble sharding over the GPUs of one node: contiguous trajectory ranges per rank, no
data-path collective; the only exchange is one all-reduce(sum) of dL/dp (np doubles) over RCCL/xGMI when p is
shared.  du0 stays sharded.  One process per GPU, torch.distributed (backend "nccl" = RCCL; "gloo" in CPU tests)."""
import numpy as np


def shard_range(n_total, rank, world_size):
    """Contiguous range [lo, hi) of trajectories owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_dp(dp, group=None):
    """Sum the per-rank dL/dp contributions.  `dp`: torch tensor (device tensor for nccl, CPU for gloo) or numpy."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dp
    if isinstance(dp, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(dp))
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    dist.all_reduce(dp, op=dist.ReduceOp.SUM, group=group)
    return dp


def gather_du0(du0_local, n_total, group=None):
    """Optional: assemble the sharded du0 [N][n] on every rank (all_gather of unequal shards via padding)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return du0_local
    ws = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, ws) for r in range(ws)]
    maxn = max(hi - lo for lo, hi in sizes)
    t = torch.as_tensor(du0_local)
    pad = torch.zeros((maxn, t.shape[1]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def comm_unique_id():
    """A fresh RCCL unique id (128 bytes) from the library (hipadj_comm_unique_id): rank 0 calls this and ships the bytes."""
    import ctypes as C
    from . import _lib
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    rc = _lib.load().hipadj_comm_unique_id(buf)
    if rc != _lib.OK:
        raise _lib.HipadjError(rc, _lib.load().hipadj_last_error(None).decode())
    return buf.raw


def init_native_allreduce(engine, group=None):
    """Gives `engine` (this rank's shard) the library's own RCCL communicator over the ranks of the torch.distributed group:
    torch.distributed only carries the 128-byte unique id from rank 0 to the others (any backend: gloo or nccl); afterwards
    every engine.adjoint / adjoint_dev all-reduces dL/dp in-stream inside the C ABI and torch is not involved in the exchange
    — the way a non-Python host (the Julia glue of INTEGRATION.md) runs the sharded path."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised (it ships the unique id to the other ranks)")
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    engine.comm_init_rank(box[0], ws, rank)
    return engine
