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
    """All-reduce the pieces of uq_stats with ONE collective: a SUM over the 512 KiB count table with 5 extra
    slots per rank appended -- every rank writes its five scalars (two unsigned minima with the sign bit flipped
    so that signed order == unsigned order, len_min, len_max, max record bytes) into its own slots and zeros
    elsewhere, so after the SUM everybody holds everybody's scalars and takes the MIN / MAX locally."""
    import torch
    if dist is None:
        dist, _, _ = _world()
    if dist is None:
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    slots = torch.zeros(world * 5, dtype=torch.int64, device=counts_i64.device)
    slots[rank * 5:rank * 5 + 5] = torch.cat([mins_u64_as_i64 ^ _SIGN, len_min.reshape(1), len_max.reshape(1), rec_max.reshape(1)])
    buf = torch.cat([counts_i64, slots])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    counts_i64.copy_(buf[:counts_i64.numel()])
    s = buf[counts_i64.numel():].reshape(world, 5)
    mins_u64_as_i64.copy_(s[:, 0:2].min(dim=0).values ^ _SIGN)
    len_min.copy_(s[:, 2].min().reshape(1)); len_max.copy_(s[:, 3].max().reshape(1)); rec_max.copy_(s[:, 4].max().reshape(1))


def allreduce_stats_inplace(t, stats_bytes, read_offset=0):
    """All-reduce the bytes of a `uq_stats` struct (a uint8 tensor on any device) over the process group,
    in place.  `read_offset` = global index of this rank's first read (bad-record indices become global)."""
    nbytes = C.sizeof(Stats)
    dist, _, world = _world()
    if dist is not None and stats_bytes.is_cuda and dist.get_backend() == 'gloo':
        host = stats_bytes.cpu()                      # gloo rehearsal: reduce on the host, copy back
        allreduce_stats_inplace(t, host, read_offset)
        stats_bytes.copy_(host)
        return
    i64 = stats_bytes[:nbytes - nbytes % 8].view(t.int64)
    counts = i64[:65536]
    mins = i64[65536:65538]
    if read_offset:
        mins.copy_(t.where(mins == -1, mins, mins + int(read_offset)))
    tail = stats_bytes[65538 * 8:65538 * 8 + 16].view(t.int32)   # len_min, len_max, max_record_bytes, reserved
    lmin = tail[0:1].to(t.int64) & 0xFFFFFFFF
    lmax = tail[1:2].to(t.int64)
    rmax = tail[2:3].to(t.int64)
    allreduce_stats_tensors(counts, mins, lmin, lmax, rmax)
    tail[0:1].copy_(lmin.to(t.int32)); tail[1:2].copy_(lmax.to(t.int32)); tail[2:3].copy_(rmax.to(t.int32))


def allreduce_stats(ctx, d_stats, read_offset=0):
    """All-reduce a device `uq_stats` over the process group and fetch it: export kernel -> ONE all-reduce SUM ->
    import kernel (uq_stats_export / uq_stats_import; allreduce_stats_inplace is the same exchange spelled in
    tensor operations, for the CPU tests)."""
    import ctypes
    from . import ops
    from ._lib import call
    dist, rank, world = _world()
    if dist is not None:
        t = ctx.torch
        buf = t.empty(65536 + 6 * world, dtype=t.int64, device=ctx.device)
        call('uq_stats_export', ctx.h, ctypes.c_void_p(d_stats.data_ptr()), rank, world, int(read_offset), ctypes.c_void_p(buf.data_ptr()))
        if dist.get_backend() == 'gloo':
            host = buf.cpu(); dist.all_reduce(host, op=dist.ReduceOp.SUM); buf.copy_(host)      # rehearsal on one card
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        call('uq_stats_import', ctx.h, ctypes.c_void_p(buf.data_ptr()), world, ctypes.c_void_p(d_stats.data_ptr()))
    return ops.stats_fetch(ctx, d_stats)


class Shard:
    """Where this rank's reads sit in the whole file, and small-control-data collectives over the group
    (python ints / bytes in, python ints / bytes out: a few hundred bytes per call)."""

    def __init__(self, be, read_offset, total, group=None):
        self.be, self.read_offset, self.total, self.group = be, int(read_offset), int(total), group
        self.dist, self.rank, self.world = _world()

    def reduce(self, values, op):
        """Element-wise all-reduce of signed 64-bit python ints; op in 'min' | 'max' | 'sum'."""
        values = [int(v) for v in values]
        if self.world == 1 or not values:
            return values
        torch, dist = self.be.torch, self.dist
        dev = self.be.device if dist.get_backend(self.group) != 'gloo' else 'cpu'
        t = torch.tensor(values, dtype=torch.int64, device=dev)
        dist.all_reduce(t, op={'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX, 'sum': dist.ReduceOp.SUM}[op], group=self.group)
        return [int(v) for v in t.cpu().tolist()]

    def gather_ints(self, value):
        """One python int per rank -> list over ranks."""
        if self.world == 1:
            return [int(value)]
        out = [0] * self.world
        out[self.rank] = int(value)
        return self.reduce(out, 'sum')

    def gather_bytes(self, data):
        """One bytes object per rank -> list over ranks."""
        if self.world == 1:
            return [bytes(data)]
        torch, dist = self.be.torch, self.dist
        lens = self.gather_ints(len(data))
        cap = max(lens) if lens else 0
        if cap == 0:
            return [b''] * self.world
        dev = self.be.device if dist.get_backend(self.group) != 'gloo' else 'cpu'
        mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
        if len(data):
            mine[:len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
        allb = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allb, mine, group=self.group)
        return [bytes(allb[r][:lens[r]].cpu().numpy().tobytes()) for r in range(self.world)]

    def gather_rows(self, rows_u8):
        """All-gather of variable-length uint8 device tensors, concatenated in rank order (stays on the device)."""
        if self.world == 1:
            return rows_u8
        torch = self.be.torch
        parts = exchange_v([rows_u8 for _ in range(self.world)], self.dist, torch, self.be.device, torch.uint8, self.group)
        return torch.cat(parts)


# --------------------------------------------------------------------------- global --sort
class HipRows:
    """Row and index operations of one rank, on the GPU through the C ABI (the product backend).  The CPU tests of the exchange
    logic pass a numpy object with the same methods (tests/test_dist_gloo.py)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.torch = ctx.torch
        self.device = ctx.device

    def argsort_rows(self, table, rows, cols):
        from . import ops
        return ops.argsort_rows(self.ctx, table, rows, cols)

    def argsort_groups(self, table, rows, cols):
        """(stable memcmp order, group id of every sorted position, number of groups): the sort's own head flags, no second look at the rows."""
        from . import ops
        perm, _, group, _, nu = ops.unique_rows(self.ctx, table, rows, cols, want_key=False, want_unique=False)
        return perm, group, nu

    def unique_rows_of_groups(self, table, rows, cols, group, nunique, perm=None):
        from . import ops
        return ops.unique_rows_of_groups(self.ctx, table, rows, cols, group, nunique, perm)

    def partition_order(self, dest, n, ndest):
        """Stable partition of positions by a one-byte destination: (order int32[n], counts int64[ndest] on the device) -- uq_partition_order."""
        from . import ops
        return ops.partition_order(self.ctx, dest, n, ndest)

    def gather_rows(self, table, rows, cols, index):
        from . import ops
        return ops.gather_rows(self.ctx, table, rows, cols, index)

    def lower_bound_rows(self, sorted_table, rows, cols, probes, nprobes):
        from . import ops
        return ops.lower_bound_rows(self.ctx, sorted_table, rows, cols, probes, nprobes)

    def partition_rows(self, splitters, nsplit, cols, table, rows, index_base, total):
        """uint8 destination rank per row (uq_partition_rows)."""
        from ._lib import call
        t = self.torch
        dest = t.empty(rows, dtype=t.uint8, device=self.device)
        call('uq_partition_rows', self.ctx.h, C.c_void_p(splitters.data_ptr()), int(nsplit), int(cols), C.c_void_p(table.data_ptr()), int(rows),
             int(index_base), int(total), C.c_void_p(dest.data_ptr()))
        return dest

    def owner_of_rows(self, gidx, starts):
        from ._lib import call
        t = self.torch
        n, world = int(gidx.numel()), len(starts) - 1
        owner = t.empty(n, dtype=t.uint8, device=self.device)
        call('uq_owner_of_rows', self.ctx.h, C.c_void_p(gidx.data_ptr()), n, (C.c_int64 * (world + 1))(*[int(x) for x in starts]), world,
             C.c_void_p(owner.data_ptr()))
        return owner

    def index_affine(self, index, add, out_itemsize):
        """index (int32 = unsigned 32-bit positions, or int64) + add -> int32 / int64 tensor (uq_index_affine)."""
        from ._lib import call
        t = self.torch
        n = int(index.numel())
        out = t.empty(n, dtype=t.int32 if out_itemsize == 4 else t.int64, device=self.device)
        call('uq_index_affine', self.ctx.h, C.c_void_p(index.data_ptr()), index.element_size(), n, int(add), C.c_void_p(out.data_ptr()), out_itemsize)
        return out

    def scatter_rows(self, values, n, cols, index, base, out_rows):
        """out[index[j] - base] = row j of `values` (uq_scatter_rows); every row of `out` must be hit exactly once by the caller's contract."""
        from ._lib import call
        t = self.torch
        out = t.empty(out_rows * cols, dtype=t.uint8, device=self.device)
        bad = C.c_uint64()
        call('uq_scatter_rows', self.ctx.h, C.c_void_p(values.data_ptr()), int(n), int(cols), C.c_void_p(index.data_ptr()), index.element_size(), int(base),
             int(out_rows), C.c_void_p(out.data_ptr()), C.byref(bad))
        if bad.value != UQ_NONE:
            raise RuntimeError('row numbers received do not cover the shard (entry %d)' % bad.value)
        return out

    def invert_permutation(self, perm, base=0):
        """inv[perm[j] - base] = j as an int32 tensor (uq_invert_permutation); raises when perm is no permutation of base .. base + n - 1."""
        from ._lib import call
        t = self.torch
        n = int(perm.numel())
        inv = t.empty(n, dtype=t.int32, device=self.device)
        bad = C.c_uint64()
        call('uq_invert_permutation', self.ctx.h, C.c_void_p(perm.data_ptr()), perm.element_size(), n, int(base), C.c_void_p(inv.data_ptr()), C.byref(bad))
        if bad.value != UQ_NONE:
            raise RuntimeError('row numbers received do not cover the shard (entry %d)' % bad.value)
        return inv


def exchange_split(send, splits, dist, torch, device, dtype, group=None, rcounts=None):
    """all-to-all(v) of ONE contiguous 1-D tensor: elements [sum(splits[:d]), sum(splits[:d + 1])) go to rank d.  Returns
    (received tensor, list of received counts per source).  `all_to_all_single` (on the nccl backend RCCL's all-to-all: every xGMI
    link carries its own peer's slice at once); when the caller does not know the receive counts (`rcounts`) they are exchanged
    first, which costs ONE host round trip (the output buffer has to be allocated)."""
    world = dist.get_world_size(group)
    out_device = device
    if dist.get_backend(group) == 'gloo' and torch.device(device).type != 'cpu':
        send = send.cpu()                            # gloo rehearsal on a GPU box (several ranks sharing one card): staged through the host
        device = 'cpu'
    splits = [int(x) for x in splits]
    if rcounts is None:
        sc = torch.tensor(splits, dtype=torch.int64, device=device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=group)
        rcounts = [int(x) for x in rc.tolist()]
    rcounts = [int(x) for x in rcounts]
    recv = torch.empty(sum(rcounts), dtype=dtype, device=device)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=rcounts, input_split_sizes=splits, group=group)
    if out_device != device:
        recv = recv.to(out_device)
    return recv, rcounts


def exchange_v(parts, dist, torch, device, dtype, group=None):
    """all-to-all(v): parts[d] (1-D tensor) goes to rank d; returns the list received from every source."""
    send = torch.cat([p.reshape(-1) for p in parts]) if len(parts) > 1 else parts[0].reshape(-1)
    recv, rcounts = exchange_split(send, [int(p.numel()) for p in parts], dist, torch, device, dtype, group)
    return list(recv.split(rcounts))


def gather_matrix(row, dist, torch, device, group=None):
    """Every rank's list of `world` ints -> the world x world matrix on every rank (one all-gather, one host read)."""
    world = dist.get_world_size(group)
    dev = 'cpu' if dist.get_backend(group) == 'gloo' else device
    mine = torch.tensor([int(x) for x in row], dtype=torch.int64, device=dev)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine, group=group)
    return [[int(x) for x in r.tolist()] for r in allr]


def gather_counts(counts_dev, dist, torch, group=None):
    """Every rank's int64[world] send counts (a device tensor, never read by the host before) -> the world x world matrix as python
    lists on every rank: ONE all-gather and ONE host read -- the only host round trip of an exchange."""
    world = dist.get_world_size(group)
    mine = counts_dev.cpu() if dist.get_backend(group) == 'gloo' else counts_dev
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine.contiguous(), group=group)
    m = torch.stack(allr).cpu().tolist()
    return [[int(x) for x in row] for row in m]


# Sample rows per rank for the splitters.  The largest shard after the exchange exceeds the mean by about 2 / sqrt(samples per rank)
# (W - 1 quantiles of W k samples: the relative spread of a rank's share is ~ sqrt(W / (W k)) ...): 1024 -> +- 6 %, 16384 -> +- 1.6 %;
# 8 x 16384 sampled rows are sorted in well under a millisecond.
SAMPLES_PER_RANK = 16384


def _sample_positions(rows, k, seed):
    """k stratified pseudo-random row numbers of a shard (host control data: a few KB)."""
    if rows == 0:
        return np.zeros(0, dtype=np.int32)
    rng = np.random.default_rng(seed)
    k = min(k, rows)
    edges = (np.arange(k + 1, dtype=np.int64) * rows) // k
    return (edges[:-1] + (rng.random(k) * (edges[1:] - edges[:-1])).astype(np.int64)).astype(np.int32)


def global_sort_rows(be, table, rows, cols, read_offset, group=None, samples_per_rank=None, total_rows=None, rows_of_ranks=None, want='sorted'):
    """Sample sort of row shards over the process group -- ONE sort per row.  Every rank passes its UNSORTED shard (`rows` x `cols`
    bytes, records [read_offset, read_offset + rows) of the file) and gets back a contiguous range of the globally sorted order:
    dict(table=sorted rows, rows=m, gidx=int64 file-wide index of each row, offset=global position of the first row,
    group=int32 group id of every row (dense ranks of the row values inside this rank's range), ngroups, edge=its first and last row).
    want='unique' (a keyed table: uq.py:784-789): instead of `table` the dict holds unique=the range's distinct rows in order -- the
    duplicates are never moved into sorted order.
      1. a stratified sample of every shard is all-gathered and sorted on the device; W - 1 splitters are its quantiles;
      2. every row's destination = the number of splitters below it (uq_partition_rows) -- no local sort is needed for that;
      3. stable partition of the shard's positions by destination (uq_partition_order: a histogram, a scan, a ranking pass -- no sort),
         rows and file indices gathered into send order, all-to-all(v) of both;
      4. ONE stable sort of what arrived.  The runs arrive in source-rank order and each is in file order, so equal rows end up in
         file order: the global order is THE stable memcmp order, whatever the splitters.  The sort's own head flags give the groups
         (nobody compares neighbouring rows again).
    Equal rows share a destination, except a value that takes up several splitters (a tie group heavier than a rank's share): it is
    dealt over those ranks by file position, so consumers that count groups must look at their neighbours' boundary rows
    (dist_encode._unique).  Host round trips: ONE (the count matrix: all-gather of the device-side send counts, read once);
    `rows_of_ranks` (every rank's shard size: callers know it from the load) spares the sample exchange its own."""
    dist, rank, world = _world()
    torch = be.torch
    if world == 1:
        if rows:
            perm, grp, ng = be.argsort_groups(table, rows, cols)
            return _sorted_range(be, table, rows, cols, perm, grp, ng, be.index_affine(perm, int(read_offset), 8), 0, want)
        return dict(table=table, unique=table, rows=0, gidx=torch.empty(0, dtype=torch.int64, device=be.device), offset=0,
                    group=torch.empty(0, dtype=torch.int32, device=be.device), ngroups=0, edge=table)
    if rows_of_ranks is None:
        rows_of_ranks = [r[0] for r in gather_matrix([rows], dist, torch, be.device, group)]
    if total_rows is None:
        total_rows = sum(rows_of_ranks)
    # 1. samples -> splitters (identical on every rank: same gathered rows, same deterministic sort)
    k = samples_per_rank or SAMPLES_PER_RANK
    ks = [min(k, int(r)) for r in rows_of_ranks]
    pick = torch.from_numpy(_sample_positions(rows, k, 20261003 + rank)).to(be.device)
    samp = be.gather_rows(table, rows, cols, pick) if rows else torch.empty(0, dtype=torch.uint8, device=be.device)
    allsamp, _ = exchange_split(torch.cat([samp for _ in range(world)]) if world > 1 else samp, [ks[rank] * cols] * world, dist, torch, be.device, torch.uint8, group,
                                rcounts=[x * cols for x in ks])
    ns = int(allsamp.numel()) // cols
    if ns:
        sorder = be.argsort_rows(allsamp, ns, cols)
        qpos = torch.tensor([min(ns - 1, (j * ns) // world) for j in range(1, world)], dtype=torch.int32, device=be.device)
        d_split = be.gather_rows(allsamp, ns, cols, be.gather_rows(sorder.view(torch.uint8), ns, 4, qpos).view(torch.int32))
    else:
        d_split = torch.zeros((world - 1) * cols, dtype=torch.uint8, device=be.device)
    # 2 + 3. destinations, stable partition, exchange
    if rows:
        dest = be.partition_rows(d_split, world - 1, cols, table, rows, int(read_offset), int(total_rows))
        order, counts = be.partition_order(dest, rows, world)
        send_rows = be.gather_rows(table, rows, cols, order)
        send_idx = be.index_affine(order, int(read_offset), 8)
    else:
        send_rows, send_idx = table, torch.empty(0, dtype=torch.int64, device=be.device)
        counts = torch.zeros(world, dtype=torch.int64, device=be.device)
    matrix = gather_counts(counts, dist, torch, group)                      # matrix[s][d] = rows rank s sends to rank d
    nsend = matrix[rank]
    nrecv = [matrix[s][rank] for s in range(world)]
    merged, _ = exchange_split(send_rows, [c * cols for c in nsend], dist, torch, be.device, torch.uint8, group, rcounts=[c * cols for c in nrecv])
    midx, _ = exchange_split(send_idx, nsend, dist, torch, be.device, torch.int64, group, rcounts=nrecv)
    m = sum(nrecv)
    # 4. the one sort
    offset = sum(sum(matrix[s][d] for s in range(world)) for d in range(rank))
    if m:
        perm2, grp, ng = be.argsort_groups(merged, m, cols)
        oidx = be.gather_rows(midx.view(torch.uint8), m, 8, perm2).view(torch.int64)
        return _sorted_range(be, merged, m, cols, perm2, grp, ng, oidx, offset, want)
    return dict(table=merged, unique=merged, rows=0, gidx=midx, offset=offset, group=torch.empty(0, dtype=torch.int32, device=be.device), ngroups=0, edge=merged)


def _sorted_range(be, table, m, cols, perm, grp, ng, gidx, offset, want):
    """What global_sort_rows hands back for a rank's range: the rows in sorted order, or (want='unique') only its distinct rows."""
    torch = be.torch
    ends = torch.stack([perm[0], perm[m - 1]])
    out = dict(rows=m, gidx=gidx, offset=offset, group=grp, ngroups=ng, edge=be.gather_rows(table, m, cols, ends))
    if want == 'unique': out['unique'] = be.unique_rows_of_groups(table, m, cols, grp, ng, perm)
    else: out['table'] = be.gather_rows(table, m, cols, perm)
    return out


def _route_by_owner(be, shard_starts, gidx, world, dist=None, group=None):
    """Group file-wide row numbers by owning rank, stably: (order, grouped gidx, send counts per destination rank, count matrix).
    The counts stay on the device until the count matrix is read (gather_counts: the exchange's one host round trip)."""
    torch = be.torch
    n = int(gidx.numel())
    if n == 0:
        counts = torch.zeros(world, dtype=torch.int64, device=be.device)
        order, sorted_idx = torch.empty(0, dtype=torch.int32, device=be.device), gidx
    else:
        owner = be.owner_of_rows(gidx, shard_starts)
        order, counts = be.partition_order(owner, n, world)
        sorted_idx = be.gather_rows(gidx.view(torch.uint8), n, 8, order).view(torch.int64)
    matrix = gather_counts(counts, dist, torch, group)
    return order, sorted_idx, matrix


def dist_scatter_rows(be, values, cols, shard_starts, gidx, group=None):
    """The inverse of dist_gather_rows: `values` (len(gidx) rows of `cols` bytes) are rows of a file-ordered
    global table at row numbers `gidx`; every rank receives the rows it owns and returns them in file order
    (uint8 tensor, rows = its shard size).  Every global row must be sent exactly once over all ranks."""
    dist, rank, world = _world()
    torch = be.torch
    n = int(gidx.numel())
    mine = int(shard_starts[rank + 1]) - int(shard_starts[rank])
    if world == 1:
        return be.scatter_rows(values, n, cols, gidx, int(shard_starts[0]), n) if n else values
    order, sorted_idx, matrix = _route_by_owner(be, shard_starts, gidx, world, dist, group)
    sorted_vals = be.gather_rows(values, n, cols, order) if n else values
    nsend, nrecv = matrix[rank], [matrix[s][rank] for s in range(world)]
    ridx, _ = exchange_split(sorted_idx, nsend, dist, torch, be.device, torch.int64, group, rcounts=nrecv)
    rval, _ = exchange_split(sorted_vals, [k * cols for k in nsend], dist, torch, be.device, torch.uint8, group, rcounts=[k * cols for k in nrecv])
    if int(ridx.numel()) != mine:
        raise RuntimeError('dist_scatter_rows: received %d rows for a shard of %d' % (int(ridx.numel()), mine))
    return be.scatter_rows(rval, mine, cols, ridx, int(shard_starts[rank]), mine) if mine else rval


def dist_gather_rows(be, table, rows, cols, shard_starts, gidx, group=None):
    """out[j] = global_table[gidx[j]] where the global table is sharded by records: rank r owns global rows
    [shard_starts[r], shard_starts[r+1]).  Requests travel to the owners, rows travel back."""
    dist, rank, world = _world()
    torch = be.torch
    n = int(gidx.numel())
    if world == 1:
        return be.gather_rows(table, rows, cols, be.index_affine(gidx, -int(shard_starts[0]), 4))
    order, sorted_idx, matrix = _route_by_owner(be, shard_starts, gidx, world, dist, group)
    nsend, nreq = matrix[rank], [matrix[s][rank] for s in range(world)]
    req, _ = exchange_split(sorted_idx, nsend, dist, torch, be.device, torch.int64, group, rcounts=nreq)
    # one gather serves all the requesters: the reply buffer is the requests' order, i.e. already grouped by destination
    want = be.index_affine(req, -int(shard_starts[rank]), 4)
    reply = be.gather_rows(table, rows, cols, want) if want.numel() else torch.empty(0, dtype=torch.uint8, device=be.device)
    got, _ = exchange_split(reply, [k * cols for k in nreq], dist, torch, be.device, torch.uint8, group, rcounts=[k * cols for k in nsend])
    # rows came back in `order`; undo it: out[order[j]] = got[j]
    return be.gather_rows(got, n, cols, be.invert_permutation(order, 0)) if n else got
