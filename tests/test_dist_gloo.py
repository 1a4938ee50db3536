"""CPU, world_size 2 and 3 over gloo: the multi-GPU exchange logic of uq_amd.dist (statistics all-reduce,
sample-sort all-to-all, distributed gather).  The row operations are supplied by a numpy backend defined
here (test infrastructure); on the GPU the same code runs with dist.HipRows."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from uq_amd import dist as uqdist


class NumpyRows:
    """CPU stand-in for dist.HipRows (stable memcmp row sort, gather, lower bound)."""
    torch = torch
    device = torch.device('cpu')

    def argsort_rows(self, table, rows, cols):
        t = table.numpy().reshape(rows, cols)
        return torch.from_numpy(np.lexsort([t[:, c] for c in range(cols - 1, -1, -1)]).astype(np.int32))

    def gather_rows(self, table, rows, cols, index):
        t = table.numpy().reshape(rows, cols)
        idx = index.numpy().astype(np.int64) & 0xFFFFFFFF if index.dtype == torch.int32 else index.numpy().astype(np.int64)
        return torch.from_numpy(np.ascontiguousarray(t[idx]).reshape(-1))

    def lower_bound_rows(self, sorted_table, rows, cols, probes, nprobes):
        t = sorted_table.numpy().reshape(rows, cols)
        p = probes.numpy().reshape(nprobes, cols)
        v = lambda a: [bytes(r) for r in a]
        import bisect
        keys = v(t)
        return torch.tensor([bisect.bisect_left(keys, k) for k in v(p)], dtype=torch.int64)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, fn, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world)))
    finally:
        dist.destroy_process_group()


def _run(world, fn):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, q)) for r in range(world)]
    for p in procs: p.start()
    out = dict(q.get() for _ in range(world))
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    return [out[r] for r in range(world)]


def _table(n, cols, nd, seed=5):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(nd, cols)).astype(np.uint8)
    return base[rng.randint(0, nd, size=n)]


def _sort_job(rank, world):
    n, cols = 5000, 13
    T = _table(n, cols, 600)
    lo, hi = uqdist.shard_range(n, rank, world)
    res = uqdist.global_sort_rows(NumpyRows(), torch.from_numpy(T[lo:hi].reshape(-1).copy()), hi - lo, cols, lo)
    # apply the order to a second table sharded the same way
    U = _table(n, 7, 5000, seed=9)
    starts = [uqdist.shard_range(n, r, world)[0] for r in range(world)] + [n]
    g = uqdist.dist_gather_rows(NumpyRows(), torch.from_numpy(U[lo:hi].reshape(-1).copy()), hi - lo, 7, starts, res['gidx'])
    return dict(rows=res['table'].numpy().reshape(-1, cols), gidx=res['gidx'].numpy(), offset=res['offset'], other=g.numpy().reshape(-1, 7))


@pytest.mark.parametrize('world', [2, 3])
def test_global_sort_and_gather(world):
    outs = _run(world, _sort_job)
    n, cols = 5000, 13
    T = _table(n, cols, 600)
    U = _table(n, 7, 5000, seed=9)
    order = np.lexsort([T[:, c] for c in range(cols - 1, -1, -1)])           # stable, memcmp order
    assert [o['offset'] for o in outs] == list(np.cumsum([0] + [len(o['gidx']) for o in outs[:-1]]))
    gidx = np.concatenate([o['gidx'] for o in outs])
    assert np.array_equal(gidx, order)
    assert np.array_equal(np.concatenate([o['rows'] for o in outs]), T[order])
    assert np.array_equal(np.concatenate([o['other'] for o in outs]), U[order])
    # equal rows never straddle ranks (unique needs no boundary fix)
    for a, b in zip(outs[:-1], outs[1:]):
        if len(a['rows']) and len(b['rows']): assert bytes(a['rows'][-1]) != bytes(b['rows'][0])


def _stats_job(rank, world):
    counts = torch.full((65536,), rank + 1, dtype=torch.int64)
    none = -1
    mins = torch.tensor([none if rank == 0 else 1000 + rank, none], dtype=torch.int64)   # UQ_NONE as int64 = -1
    lmin = torch.tensor([100 + rank], dtype=torch.int64); lmax = torch.tensor([150 - rank], dtype=torch.int64)
    rmax = torch.tensor([340 + rank], dtype=torch.int64)
    uqdist.allreduce_stats_tensors(counts, mins, lmin, lmax, rmax)
    return dict(c=int(counts[0]), mins=mins.tolist(), lmin=int(lmin), lmax=int(lmax), rmax=int(rmax))


@pytest.mark.parametrize('world', [2])
def test_allreduce_stats(world):
    outs = _run(world, _stats_job)
    for o in outs:
        assert o['c'] == 3 and o['mins'] == [1001, -1] and o['lmin'] == 100 and o['lmax'] == 150 and o['rmax'] == 341


def _stats_struct_job(rank, world):
    import ctypes
    from uq_amd._lib import Stats
    s = Stats()
    s.counts[65 * 256 + 73] = 10 * (rank + 1); s.counts[255 * 256 + 255] = 1
    s.bad_plus = (1 << 64) - 1 if rank == 0 else 7
    s.bad_len = (1 << 64) - 1
    s.len_min = 0xFFFFFFFF if rank == 1 else 36; s.len_max = 100 + rank; s.max_record_bytes = 500 - rank
    buf = torch.frombuffer(bytearray(bytes(s)), dtype=torch.uint8).clone()
    uqdist.allreduce_stats_inplace(torch, buf, read_offset=1000 * rank)
    out = Stats.from_buffer_copy(buf.numpy().tobytes())
    return dict(c=int(out.counts[65 * 256 + 73]), c2=int(out.counts[255 * 256 + 255]), bp=int(out.bad_plus), bl=int(out.bad_len),
                lmin=int(out.len_min), lmax=int(out.len_max), rmax=int(out.max_record_bytes))


def test_allreduce_stats_struct():
    outs = _run(2, _stats_struct_job)
    for o in outs:
        assert o == dict(c=30, c2=2, bp=1007, bl=(1 << 64) - 1, lmin=36, lmax=101, rmax=500)


def test_shard_ranges_cover():
    for total in (0, 1, 7, 1000):
        for world in (1, 2, 3, 8):
            r = [uqdist.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
