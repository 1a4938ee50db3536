"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Reads shard record-parallel: rank r owns a contiguous range of records.  Without --sort the only
exchange is the all-reduce of the pass-1 statistics (so that every rank derives the same alphabets and
bit widths); the packed tables of the ranks are then just concatenated in rank order.  The global
--sort exchange (sample sort, all-to-all(v) of rows by key range) lives in `global_sort`.

Nothing here computes on table bytes: compute goes through the C ABI (or, in the CPU tests of the
exchange logic, through the backend object the test injects).
"""
import ctypes as C

import numpy as np

from ._lib import Stats, UQ_NONE

_SIGN = -(1 << 63)


def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def shard_range(total, rank, world):
    """Contiguous record range of `rank`: [lo, hi)."""
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi


def allreduce_stats_tensors(counts_i64, mins_u64_as_i64, len_min, len_max, rec_max, dist=None):
    """All-reduce the pieces of uq_stats.  `counts_i64`: SUM.  `mins_u64_as_i64`: unsigned MIN (done as
    a signed MIN after flipping the sign bit, which preserves unsigned order).  Scalars: MIN / MAX."""
    import torch
    if dist is None:
        dist, _, _ = _world()
    if dist is None:
        return
    dist.all_reduce(counts_i64, op=dist.ReduceOp.SUM)
    mins_u64_as_i64 ^= _SIGN
    dist.all_reduce(mins_u64_as_i64, op=dist.ReduceOp.MIN)
    mins_u64_as_i64 ^= _SIGN
    dist.all_reduce(len_min, op=dist.ReduceOp.MIN)
    mx = torch.stack([len_max.reshape(()), rec_max.reshape(())])
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    len_max.copy_(mx[0]); rec_max.copy_(mx[1])


def allreduce_stats(ctx, d_stats, read_offset=0):
    """All-reduce a device `uq_stats` over the process group and fetch it.  `read_offset` = global index
    of this rank's first read (so that bad-record indices are global)."""
    from . import ops
    t = ctx.torch
    nbytes = C.sizeof(Stats)
    i64 = d_stats[:nbytes - nbytes % 8].view(t.int64)
    counts = i64[:65536]
    mins = i64[65536:65538]
    if read_offset:
        none = t.tensor(-1, dtype=t.int64, device=ctx.device)
        mins.copy_(t.where(mins == none, mins, mins + read_offset))
    tail = d_stats[65538 * 8:65538 * 8 + 16].view(t.int32)       # len_min, len_max, max_record_bytes, reserved
    lmin = tail[0:1].to(t.int64) & 0xFFFFFFFF
    lmax = tail[1:2].to(t.int64)
    rmax = tail[2:3].to(t.int64)
    allreduce_stats_tensors(counts, mins, lmin, lmax, rmax)
    tail[0:1].copy_(lmin.to(t.int32)); tail[1:2].copy_(lmax.to(t.int32)); tail[2:3].copy_(rmax.to(t.int32))
    return ops.stats_fetch(ctx, d_stats)
