"""CPU, world_size 2 and 3 over gloo: the multi-GPU exchange logic of uq_amd.dist (statistics all-reduce,
sample-sort all-to-all, distributed gather).  The row operations are supplied by a numpy backend defined
here (test infrastructure); on the GPU the same code runs with dist.HipRows."""
import json
import os
import socket
import time

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

    def argsort_groups(self, table, rows, cols):
        perm = self.argsort_rows(table, rows, cols)
        t = table.numpy().reshape(rows, cols)[perm.numpy().astype(np.int64)]
        head = np.ones(rows, dtype=bool); head[1:] = (t[1:] != t[:-1]).any(axis=1)
        return perm, torch.from_numpy((np.cumsum(head) - 1).astype(np.int32)), int(head.sum())

    def unique_rows_of_groups(self, table, rows, cols, group, nunique, perm=None):
        t = table.numpy().reshape(rows, cols)
        if perm is not None: t = t[perm.numpy().astype(np.int64) & 0xFFFFFFFF]
        g = group.numpy().astype(np.int64)
        first = np.ones(rows, dtype=bool); first[1:] = g[1:] != g[:-1]
        assert int(first.sum()) == nunique
        return torch.from_numpy(np.ascontiguousarray(t[first]).reshape(-1))

    def partition_order(self, dest, n, ndest):
        d = dest.numpy()[:n]
        assert d.max(initial=0) < ndest
        return torch.from_numpy(np.argsort(d, kind='stable').astype(np.int32)), torch.from_numpy(np.bincount(d, minlength=ndest).astype(np.int64))

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

    # the routing arithmetic of csrc/route.hip, restated (include/uqhip.h: uq_partition_rows, uq_owner_of_rows, uq_index_affine,
    # uq_invert_permutation)
    def partition_rows(self, splitters, nsplit, cols, table, rows, index_base, total):
        import bisect
        keys = [bytes(r) for r in splitters.numpy().reshape(nsplit, cols)]
        out = np.zeros(rows, dtype=np.uint8)
        for r, row in enumerate(table.numpy().reshape(rows, cols)):
            lb, ub = bisect.bisect_left(keys, bytes(row)), bisect.bisect_right(keys, bytes(row))
            e = ub - lb
            out[r] = lb + ((index_base + r) * e) // total if e >= 2 else lb
        return torch.from_numpy(out)

    def owner_of_rows(self, gidx, starts):
        s = np.asarray(list(starts), dtype=np.int64)
        return torch.from_numpy((np.searchsorted(s[1:-1], gidx.numpy(), side='right')).astype(np.uint8))

    def index_affine(self, index, add, out_itemsize):
        a = index.numpy().astype(np.int64)
        if index.dtype == torch.int32: a &= 0xFFFFFFFF
        return torch.from_numpy((a + add).astype(np.int32 if out_itemsize == 4 else np.int64))

    def scatter_rows(self, values, n, cols, index, base, out_rows):
        a = index.numpy().astype(np.int64)
        if index.dtype == torch.int32: a &= 0xFFFFFFFF
        a = a - base
        assert sorted(a.tolist()) == list(range(out_rows))
        out = np.empty((out_rows, cols), dtype=np.uint8)
        out[a] = values.numpy().reshape(n, cols)
        return torch.from_numpy(out.reshape(-1))

    def invert_permutation(self, perm, base=0):
        a = perm.numpy().astype(np.int64)
        if perm.dtype == torch.int32: a &= 0xFFFFFFFF
        a = a - base
        assert sorted(a.tolist()) == list(range(len(a)))
        inv = np.empty(len(a), dtype=np.int32)
        inv[a] = np.arange(len(a), dtype=np.int32)
        return torch.from_numpy(inv)


class _Failed:
    def __init__(self, text): self.text = text


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, fn, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, fn(rank, world)))
    except BaseException as e:                   # a rank that fails says so: the parent must not wait for its result
        import traceback
        q.put((rank, _Failed(''.join(traceback.format_exception(type(e), e, e.__traceback__)))))
        os._exit(1)                              # peers stuck in a collective with this rank get 'connection closed'
    finally:
        dist.destroy_process_group()


def _run(world, fn):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, q)) for r in range(world)]
    for p in procs: p.start()
    out = {}
    deadline = time.time() + 120
    while len(out) < world and time.time() < deadline:
        if not q.empty():
            r, v = q.get(); out[r] = v
        elif any(p.exitcode not in (None, 0) for p in procs) and q.empty():
            time.sleep(0.5)                      # a rank died: collect what is queued, then stop waiting
            while not q.empty():
                r, v = q.get(); out[r] = v
            break
        else:
            time.sleep(0.02)
    for p in procs: p.join(10)
    for p in procs:
        if p.is_alive(): p.kill()
    failed = [v.text for v in out.values() if isinstance(v, _Failed)]
    assert not failed, failed[0]
    assert len(out) == world and all(p.exitcode == 0 for p in procs), 'ranks %s did not finish' % [r for r in range(world) if r not in out]
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
    # with 600 distinct rows and 2-3 ranks no value is heavier than a rank's share: equal rows do not straddle ranks
    for a, b in zip(outs[:-1], outs[1:]):
        if len(a['rows']) and len(b['rows']): assert bytes(a['rows'][-1]) != bytes(b['rows'][0])


def _skew_job(rank, world):
    n, cols = 6000, 9
    T = _table(n, cols, 50, seed=11)
    T[np.random.RandomState(3).rand(n) < 0.7] = T[0]                       # one value holds 70 % of the rows
    lo, hi = uqdist.shard_range(n, rank, world)
    res = uqdist.global_sort_rows(NumpyRows(), torch.from_numpy(T[lo:hi].reshape(-1).copy()), hi - lo, cols, lo, total_rows=n)
    return dict(rows=res['table'].numpy().reshape(-1, cols), gidx=res['gidx'].numpy(), offset=res['offset'])


@pytest.mark.parametrize('world', [3, 4])
def test_global_sort_deals_a_heavy_tie_group_over_ranks(world):
    """A value heavier than a rank's share takes up several splitters; its rows are dealt to those ranks by file position: the
    global order stays THE stable order and no rank ends up with (nearly) everything."""
    outs = _run(world, _skew_job)
    n, cols = 6000, 9
    T = _table(n, cols, 50, seed=11)
    T[np.random.RandomState(3).rand(n) < 0.7] = T[0]
    order = np.lexsort([T[:, c] for c in range(cols - 1, -1, -1)])
    assert np.array_equal(np.concatenate([o['gidx'] for o in outs]), order)
    assert np.array_equal(np.concatenate([o['rows'] for o in outs]), T[order])
    assert [o['offset'] for o in outs] == list(np.cumsum([0] + [len(o['gidx']) for o in outs[:-1]]))
    assert max(len(o['gidx']) for o in outs) < 0.6 * n, [len(o['gidx']) for o in outs]


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


def _qname_cases():
    from uq_amd import synth
    cases = {'illumina': synth.fastq(6, 25000, 8)}
    # checkpoint demotions at 10 000 / 20 000 reads land in different shards; a mapping column of strings
    names = []
    for i in range(23000):
        b = i % 1000 if i <= 10000 else i % 2001
        names.append(b'@q:%d:%d:%s/%d' % ((i % 1001) * 3, b * 1000, [b'x', b'yy', b'zzz'][i % 3], 1 + i % 2))
    cases['checkpoints'] = b''.join(n + b'\nACGT\n+\nIIII\n' for n in names)
    cases['strings'] = b''.join(b'@q:%d:%s\nA\n+\nI\n' % (i % 7, b'abc' if i == 11000 else b'%d' % i) for i in range(12000))   # refused
    # heavy ties: a string value that is heavier than a rank's share is dealt over several ranks by the global sort; the
    # distinct counts must count it once (a filter flag N / Y at 90 %, and a constant string in mid-name)
    cases['heavy_flag'] = b''.join(b'@q:%d:%s:%d\nA\n+\nI\n' % (i % 40, b'Y' if i % 10 == 3 else b'N', i % 9) for i in range(3000))
    cases['heavy_const'] = b''.join(b'@q:%d:%s:lane:%s\nA\n+\nI\n' % (i, [b'a', b'b', b'c'][(i * i) % 3], b'x' if i < 2990 else b'w') for i in range(3000))
    cases['nosep'] = b''.join(b'@r%d\nA\n+\nI\n' % (i % 10) for i in range(50))                                               # refused (Q13)
    return cases


def _qname_job(rank, world):
    import fake_qname_ops as F
    import oracle_c
    from uq_amd import qname, qname_device
    qname_device.ops = F
    out = {}
    for name, fq in _qname_cases().items():
        host = np.frombuffer(fq, dtype=np.uint8)
        ls = oracle_c.index_lines(host)
        n = (len(ls) - 1) // 4
        lo, hi = uqdist.shard_range(n, rank, world)
        b0, b1 = int(ls[4 * lo]), int(ls[4 * hi])
        buf = torch.from_numpy(host[b0:b1].copy())
        lls = torch.from_numpy((ls[4 * lo:4 * hi + 1] - b0).astype(np.uint64).view(np.int64).copy())
        shard = uqdist.Shard(NumpyRows(), lo, n)
        try:
            res = qname_device.analyse_device(F.FakeCtx(), buf, lls, hi - lo, shard)
        except qname.QnameError as e:
            out[name] = ('error', str(e)); continue
        if res is None:
            out[name] = ('declined',); continue
        pre, suf, sep, cols, arrs = res
        out[name] = ('ok', pre, suf, sep, cols, [a.numpy().view(np.dtype(c['dtype'])).copy() for a, c in zip(arrs, cols)])
    return out


@pytest.mark.parametrize('world', [2, 3, 4])
def test_sharded_qname_passes_match_oracle(world):
    """The device QNAME path over shards (MIN/MAX/SUM of its reductions + a distributed sort of the field keys for
    the distinct counts) reaches the single-process oracle's answer -- or its refusal -- on every rank."""
    import uq_oracle as O
    outs = _run(world, _qname_job)
    for name, fq in _qname_cases().items():
        lines = O.read_lines(fq)
        try:
            p1 = O.pass1(lines)
            ocols = O.qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
            oarr = O.qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], ocols)
        except O.UqError:
            assert all(o[name][0] == 'error' for o in outs), name
            continue
        assert all(o[name][0] == 'ok' for o in outs), (name, [o[name][:1] for o in outs])
        for o in outs:
            assert o[name][1:4] == (p1['prefix'], p1['suffix'], p1['separators']) and o[name][4] == ocols
        for c in range(len(ocols)):
            got = np.concatenate([o[name][5][c] for o in outs])
            assert got.dtype == oarr[c].dtype and np.array_equal(got, oarr[c]), (name, c)


def test_shard_ranges_cover():
    for total in (0, 1, 7, 1000):
        for world in (1, 2, 3, 8):
            r = [uqdist.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))


def _decode_job(rank, world, enc, out):
    import fake_decode_ops as F
    from uq_amd import dist_encode, uq
    args = uq.validate_args(uq.build_parser().parse_args(['-i', enc, '-o', out, '--decode', '--quiet']))
    s = dist_encode.ShardedSession(args, ctx=F.FakeCtx())
    s.io, s.ops = F.FakeIO(), F.FakeOps()
    s.decode_sharded(out)
    return True


@pytest.mark.parametrize('world,flags', [(2, {}), (3, dict(sort='QUAL', raw=['DNA', 'QNAME'], pattern=['2.2', '1.2'])),
                                         (2, dict(raw=['DNA', 'QUAL', 'QNAME'])), (4, dict(sort='DNA', pattern=['0.2', '3.1']))],
                         ids=['keyed', 'sorted-mixed-patterns', 'raw', 'sorted-keyed'])
def test_sharded_decoder_ranges_and_offsets(tmp_path, world, flags):
    """uq_amd.dist_encode.decode_sharded with the device layer replaced by numpy / the oracle: every rank reads its
    slices of an oracle-written container, and the file the ranks write together is the oracle's decode."""
    import functools
    import io as _io
    import tarfile
    import uq_oracle as O
    from uq_amd import synth
    fq = synth.fastq(20261003 + 42, 700, (20, 45), n_rate=2, dup='both', dup_templates=25)
    cfg, members, _ = O.encode(fq, **flags)
    enc = str(tmp_path / 'x.uQ')
    with tarfile.open(enc, 'w') as t:
        for name, data in [('config.json', json.dumps(cfg).encode())] + sorted(members.items()):
            ti = tarfile.TarInfo(name); ti.size = len(data); t.addfile(ti, _io.BytesIO(data))
    out = str(tmp_path / 'back.fastq')
    assert all(_run(world, functools.partial(_decode_job, enc=enc, out=out)))
    assert open(out, 'rb').read().decode('latin-1') == O.decode(cfg, members)


def _load_job(rank, world, path, slack):
    """ShardedSession.load over `world` ranks with the device layer replaced by numpy: returns what this rank raised."""
    import fake_decode_ops as F
    from uq_amd import dist_encode, uq
    dist_encode.SLACK = slack
    args = uq.validate_args(uq.build_parser().parse_args(['-i', path, '--quiet', '--multi-pass']))      # (the numpy stand-ins cover the separate passes; the queued fused step is the GPU suite's)
    s = dist_encode.ShardedSession(args, ctx=F.FakeCtx())
    s.io, s.ops = F.FakeIO(), F.FakeLoadOps()
    try:
        s.load(path)
    except uq.UqError as e:
        return 'UqError: ' + str(e)
    return 'ok %d reads from %d, %d in file' % (s.total, s.read_offset, s.total_reads)


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_load_errors_are_collective(tmp_path, world):
    """A failure only ONE rank can see must come out of EVERY rank as the same UqError, promptly -- not as one rank
    leaving while its peers wait in the next collective (dist_encode.py:83)."""
    import functools
    from uq_amd import synth
    fq = synth.fastq(20261003 + 44, 64, 40)
    p = tmp_path / 'x.fastq'
    # (1) intact file: every rank loads its share
    p.write_bytes(fq)
    res = _run(world, functools.partial(_load_job, path=str(p), slack=4 << 20))
    assert all(r.startswith('ok') for r in res) and sum(int(r.split()[1]) for r in res) == 64, res
    # (2) no final newline: `wc -l` counts 255 lines (uq.py:85-87) -- the single-GPU CLI's message, from every rank
    p.write_bytes(fq[:-1])
    res = _run(world, functools.partial(_load_job, path=str(p), slack=4 << 20))
    assert len(set(res)) == 1 and 'contains255rows, which is not divisible by 4' in res[0], res
    # (3) an unterminated FIFTH line after complete records: the tail is ignored exactly as `wc -l` / 4 ignores it
    p.write_bytes(fq + b'@trailing garbage')
    res = _run(world, functools.partial(_load_job, path=str(p), slack=4 << 20))
    assert all(r.startswith('ok') for r in res) and sum(int(r.split()[1]) for r in res) == 64, res
    # (4) a record longer than the slack straddles a shard boundary: only the rank in front of it can tell
    res = _run(world, functools.partial(_load_job, path=str(p), slack=16))
    assert len(set(res)) == 1 and 'straddles a shard boundary' in res[0], res
