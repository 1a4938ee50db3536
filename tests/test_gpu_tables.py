"""GPU parity for the table-build and layout kernels through the C ABI: the eight --pattern layouts and
their inverses, row argsort, gather, unique/inverse, key narrowing, QNAME column stacking, unpack.
Oracle = numpy as the reference uses it (oracle/uq_oracle.py) + the closed forms of oracle/uq_oracle.c.
Bit-exact."""
import numpy as np
import pytest

import oracle_c
import uq_oracle as O
from uq_amd import ops

pytestmark = pytest.mark.gpu


def test_hostio_staging_chunk_boundaries(ctx, tmp_path):
    """uq_amd.hostio with tiny chunks: every path crosses many chunk / buffer-reuse boundaries."""
    import io
    import os
    from uq_amd.hostio import Staging
    rng = np.random.default_rng(3)
    data = rng.integers(0, 256, 1_000_003, dtype=np.uint8)
    path = tmp_path / 'blob.bin'
    data.tofile(path)
    for chunk, nbuf in ((1 << 16, 3), (4096, 2), (1 << 20, 4), (777 * 16, 5)):
        io_ = Staging(ctx, chunk=chunk, nbuf=nbuf)
        for off, size in ((0, None), (1, 999_999), (65_535, 65_538), (1_000_002, 1), (5, 0)):
            d = io_.file_to_device(str(path), off, size)
            want = data[off:] if size is None else data[off:off + size]
            assert np.array_equal(ctx.to_numpy(d), want)
        d = ctx.to_device(data)
        out = tmp_path / ('out_%d.bin' % chunk)
        with open(out, 'wb') as f:
            f.write(b'head')
            assert io_.device_to_stream(d, f) == data.size          # seekable file: pwrite path, then the position moves on
            f.write(b'tail')
        assert out.read_bytes() == b'head' + data.tobytes() + b'tail'
        buf = io.BytesIO()
        assert io_.device_to_stream(d[7:123_456], buf) == 123_449  # no descriptor: sequential path
        assert buf.getvalue() == data[7:123_456].tobytes()
        assert np.array_equal(io_.to_numpy(d[3:]), data[3:])
        fd = os.open(str(out), os.O_WRONLY)
        try:
            assert io_.device_to_fd(d[:100_000], fd, 10) == 100_000
        finally:
            os.close(fd)
        assert out.read_bytes()[10:100_010] == data[:100_000].tobytes()


def _dev(ctx, a):
    return ctx.to_device(np.ascontiguousarray(a))


SHAPES = [(1, 1), (1, 7), (5, 1), (4, 3), (64, 38), (1000, 38), (4097, 113), (333, 227), (3000, 25), (70000, 14), (2, 300), (17, 16), (1025, 4)]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: '%dx%d' % s)
@pytest.mark.parametrize('pat', O.PATTERNS)
def test_pattern_payload_and_inverse(ctx, shape, pat):
    R, C = shape
    rng = np.random.RandomState(R * 131 + C)
    T = rng.randint(0, 256, size=(R, C)).astype(np.uint8)
    # numpy.save writes the array in its own memory order: compare with the bytes after the header
    npy = O.write_pattern(T, pat)
    payload = np.frombuffer(npy, dtype=np.uint8)[-R * C:]
    assert np.array_equal(payload, oracle_c.pattern(T, ops.PATTERN_IDS[pat]))   # closed form == numpy
    d_T = _dev(ctx, T.ravel())
    out = ops.pattern(ctx, d_T, R, C, pat)
    assert np.array_equal(ctx.to_numpy(out), payload)
    back = ops.unpattern(ctx, out, R, C, pat)
    assert np.array_equal(ctx.to_numpy(back).reshape(R, C), T)


def test_pattern_misaligned_buffers(ctx):
    R, C = 777, 38
    rng = np.random.RandomState(3)
    T = rng.randint(0, 256, size=(R, C)).astype(np.uint8)
    for off_in, off_out in [(1, 0), (0, 3), (5, 9), (15, 2)]:
        back_in = ctx.empty(R * C + 32); back_out = ctx.empty(R * C + 32)
        d_T = back_in[off_in:off_in + R * C]; d_T.copy_(ctx.torch.from_numpy(T.ravel()))
        d_o = back_out[off_out:off_out + R * C]
        for pat in O.PATTERNS:
            ops.pattern(ctx, d_T, R, C, pat, out=d_o)
            assert np.array_equal(ctx.to_numpy(d_o), oracle_c.pattern(T, ops.PATTERN_IDS[pat])), pat
            back = ops.unpattern(ctx, d_o, R, C, pat)
            assert np.array_equal(ctx.to_numpy(back).reshape(R, C), T), pat


def _rows_with_dups(rng, n, C, ndistinct, prefix=0):
    """n rows drawn from `ndistinct` distinct random rows; optional long common prefix."""
    base = rng.randint(0, 256, size=(ndistinct, C)).astype(np.uint8)
    if prefix:
        base[:, :prefix] = base[0, :prefix]
    # make some rows differ only in the LAST byte / only inside a middle chunk
    if ndistinct > 4 and C > 9:
        base[1] = base[0]; base[1, -1] ^= 1
        base[2] = base[0]; base[2, 8] ^= 0x80
    return base[rng.randint(0, ndistinct, size=n)]


SORT_CASES = [(1, 5, 1, 0), (2, 1, 2, 0), (1000, 1, 7, 0), (5000, 8, 5000, 0), (5000, 38, 300, 0), (20000, 38, 20000, 0),
              (9000, 113, 500, 40), (3000, 227, 50, 100), (4096, 16, 16, 8), (10000, 9, 3, 0), (70000, 14, 60000, 0)]


@pytest.mark.parametrize('n,C,nd,prefix', SORT_CASES, ids=lambda v: str(v))
def test_argsort_rows_stable(ctx, n, C, nd, prefix):
    rng = np.random.RandomState(n + C)
    T = _rows_with_dups(rng, n, C, nd, prefix)
    perm = ops.argsort_rows(ctx, _dev(ctx, T.ravel()), n, C)
    got = ctx.to_numpy(perm, np.uint32).astype(np.int64)
    assert np.array_equal(got, O.argsort_rows(T))


@pytest.mark.parametrize('form', ['lsd', 'msd'])
@pytest.mark.parametrize('C', [12, 38, 113])
@pytest.mark.parametrize('shape', ['constant-lead', 'lead-the-sample-misses', 'crowded', 'crowded-behind-the-sample', 'two-values', 'sorted-input'])
def test_argsort_round0_paths(ctx, shape, C, form):
    """The 32-bit round 0 of the row sort (tables of 65 536 rows and more, rows wider than 8 bytes) takes the number of constant leading bits and
    the crowding of the prefixes from a sample of 4096 rows and checks the former on every row: tables whose leading bits are constant, whose only
    rows with another leading bit the sample misses (the pass runs again with the true count), that crowd on few prefixes (64-bit round 0 at once),
    that crowd but look spread in the sample (64-bit round 0 after the 32-bit pass), of two row values, and already sorted ones."""
    rng = np.random.RandomState(len(shape) * 131 + C)
    n = 100_003
    T = rng.randint(0, 256, size=(n, C)).astype(np.uint8)
    sampled = set(int(j * (n - 1) // 4095) for j in range(4096))
    ops.sort_config(ctx, msd_min_rows=1 if form == 'msd' else -1)      # (the MSD partition checks z the same way and hands crowded tables back)
    if shape == 'constant-lead':
        T[:, 0] = 0x00; T[:, 1] = (T[:, 1] & 0x0F) | 0x30
    elif shape == 'lead-the-sample-misses':
        T[:, 0] = 0x00; T[:, 1] &= 0x0F
        odd = [i for i in (12345, 77777, 3, n - 2) if i not in sampled]
        assert odd
        T[odd, 0] = 0x80
    elif shape == 'crowded':
        heads = rng.randint(0, 256, size=(50, 8)).astype(np.uint8)
        T[:, :8] = heads[rng.randint(0, 50, size=n)]
    elif shape == 'crowded-behind-the-sample':
        # every sampled row has a prefix of its own, everything else crowds on three
        heads = rng.randint(0, 256, size=(3, 8)).astype(np.uint8)
        rest = np.array([i for i in range(n) if i not in sampled])
        T[rest, :8] = heads[rng.randint(0, 3, size=len(rest))]
    elif shape == 'two-values':
        two = rng.randint(0, 256, size=(2, C)).astype(np.uint8)
        two[1, :C - 1] = two[0, :C - 1]; two[1, C - 1] ^= 1
        T = two[rng.randint(0, 2, size=n)]
    else:
        T = T[O.argsort_rows(T)]
    perm = ops.argsort_rows(ctx, _dev(ctx, T.ravel()), n, C)
    assert np.array_equal(ctx.to_numpy(perm, np.uint32).astype(np.int64), O.argsort_rows(T))
    p2, key, skey, uniq, nu = ops.unique_rows(ctx, _dev(ctx, T.ravel()), n, C)
    ops.sort_config(ctx)
    ru, rkey = O.unique_rows(T)
    assert nu == len(ru) and np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), ru) and np.array_equal(ctx.to_numpy(key, np.uint32).astype(np.int64), rkey)


@pytest.mark.parametrize('n,C,nd,prefix', SORT_CASES, ids=lambda v: str(v))
def test_unique_rows(ctx, n, C, nd, prefix):
    rng = np.random.RandomState(n * 3 + C)
    T = _rows_with_dups(rng, n, C, nd, prefix)
    perm, key, skey, uniq, nu = ops.unique_rows(ctx, _dev(ctx, T.ravel()), n, C)
    ru, rkey = O.unique_rows(T)
    assert nu == len(ru)
    assert np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), ru)
    k = ctx.to_numpy(key, np.uint32).astype(np.int64)
    assert np.array_equal(k, rkey)
    order = np.argsort(rkey, kind='stable')
    assert np.array_equal(ctx.to_numpy(perm, np.uint32).astype(np.int64), order)
    assert np.array_equal(ctx.to_numpy(skey, np.uint32).astype(np.int64), rkey[order])
    # narrowing (uq.py:790)
    isz = ops.key_itemsize(nu - 1)
    assert isz == O.narrow_key(rkey).dtype.itemsize
    nk = ops.narrow(ctx, key, isz)
    assert np.array_equal(ctx.to_numpy(nk, O.narrow_key(rkey).dtype), O.narrow_key(rkey))


@pytest.mark.parametrize('n,C,nd,prefix', SORT_CASES, ids=lambda v: str(v))
def test_unique_sorted_rows(ctx, n, C, nd, prefix):
    """uq_unique_sorted_rows (what a rank runs on the shard the global sort handed it): on a table already in memcmp order the group
    ids and the distinct rows are numpy.unique's."""
    rng = np.random.RandomState(n * 5 + C)
    T = _rows_with_dups(rng, n, C, nd, prefix)
    T = T[O.argsort_rows(T)]
    group, uniq, nu = ops.unique_sorted_rows(ctx, _dev(ctx, T.ravel()), n, C)
    ru, rkey = O.unique_rows(T)
    assert nu == len(ru)
    assert np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), ru)
    assert np.array_equal(ctx.to_numpy(group, np.uint32).astype(np.int64), rkey)
    g2, none, nu2 = ops.unique_sorted_rows(ctx, _dev(ctx, T.ravel()), n, C, want_unique=False)
    assert none is None and nu2 == nu and ctx.torch.equal(g2, group)


@pytest.fixture
def msd(ctx):
    """Round 0 of every row sort as the MSD partition (csrc/msd.hip) whatever the row count -- the product takes it from 2^18 rows --, so that
    the shapes of this file go through it; yields a function that says how many sorts did since the last call."""
    ops.sort_config(ctx, msd_min_rows=1)
    last = [ops.sort_counters(ctx)[0]]

    def took():
        now = ops.sort_counters(ctx)[0]
        d, last[0] = now - last[0], now
        return d
    yield took
    ops.sort_config(ctx)


@pytest.mark.parametrize('n,C,nd,prefix', SORT_CASES, ids=lambda v: str(v))
def test_msd_round0_small_shapes(ctx, msd, n, C, nd, prefix):
    """The shapes of test_argsort_rows_stable / test_unique_rows through the MSD partition: order, stability, unique, inverse == numpy.
    Tables of few distinct heads are handed back to the LSD passes (the counters say which way a sort went); the answer is the same."""
    rng = np.random.RandomState(n + 7 * C)
    T = _rows_with_dups(rng, n, C, nd, prefix)
    d_T = _dev(ctx, T.ravel())
    perm = ctx.to_numpy(ops.argsort_rows(ctx, d_T, n, C), np.uint32).astype(np.int64)
    assert np.array_equal(perm, O.argsort_rows(T))
    went = msd()
    spread = nd >= 5000 and prefix == 0
    assert went == (1 if spread else went), 'a table of %d distinct random rows did not take the MSD partition' % nd
    perm2, key, skey, uniq, nu = ops.unique_rows(ctx, d_T, n, C)
    ru, rkey = O.unique_rows(T)
    assert nu == len(ru) and np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), ru)
    assert np.array_equal(ctx.to_numpy(key, np.uint32).astype(np.int64), rkey)
    assert np.array_equal(ctx.to_numpy(perm2, np.uint32).astype(np.int64), np.argsort(rkey, kind='stable'))


MSD_CASES = [
    # n, C, kind, level bits (None: the plan for n)
    (300_000, 38, 'random', None), (300_000, 113, 'random', None), (1_000_003, 38, 'dups', None), (1_000_003, 38, 'dups', [8, 4]),
    (600_000, 113, 'qual', None), (600_000, 113, 'qual', [8, 5]), (524_288, 8, 'random', None), (400_000, 4, 'random', None),
    (400_000, 6, 'dups', None), (300_000, 38, 'lead', None), (700_000, 38, 'collide32', None), (700_000, 16, 'dups', [6, 6]),
    (300_000, 38, 'random', [10, 2]), (300_000, 38, 'random', [3, 3, 3, 3]), (2_100_000, 38, 'dups', None), (300_000, 38, 'heavy', None),
    (300_000, 12, 'steps', None), (1_300_000, 16, 'dups', None), (1_100_000, 113, 'qual', None), (1_048_576, 9, 'heavy', None),
]


@pytest.mark.parametrize('n,C,kind,bits', MSD_CASES, ids=lambda v: str(v).replace(' ', ''))
def test_msd_round0(ctx, n, C, kind, bits):
    """Row sort / unique at the sizes where round 0 is the MSD partition (csrc/msd.hip): random DNA-like rows, a tenth of them copies of n / 16
    templates (configs[2]'s rule), QUAL-like rows (6-bit symbols from 41 values: crowded heads, 64-bit keys), 8-byte and narrower rows (the head
    IS the row), constant leading bits, rows that collide on their 32-bit prefix and differ behind, level plans of one to four levels, one value
    heavier than a chunk (handed back to the LSD passes), heads in few steps (QNAME-like).  Against numpy's stable order."""
    rng = np.random.RandomState(n % 1000 + C)
    if kind == 'qual':
        sym = rng.randint(0, 41, size=(n, (C * 8) // 6 + 1)).astype(np.uint8)
        bitsarr = ((sym[:, :, None] >> np.arange(5, -1, -1)) & 1).astype(np.uint8).reshape(n, -1)[:, :C * 8]
        T = np.packbits(bitsarr, axis=1)
    else:
        T = rng.randint(0, 256, size=(n, C)).astype(np.uint8)
    if kind in ('dups', 'collide32'):
        nt = max(1, n // 16)
        tmpl = rng.randint(0, 256, size=(nt, C)).astype(np.uint8)
        pick = rng.rand(n) < 0.1
        T[pick] = tmpl[rng.randint(0, nt, size=int(pick.sum()))]
    if kind == 'collide32' and C > 9:
        # pairs and triples that agree on their first 6 bytes and differ behind: in the last byte, right behind the head, in the middle
        src = rng.randint(0, n, size=n // 20)
        dst = rng.randint(0, n, size=n // 20)
        T[dst, :6] = T[src, :6]
        T[dst[::3], 6:] = T[src[::3], 6:]; T[dst[::3], -1] ^= 1
    if kind == 'lead':
        T[:, 0] = 0; T[:, 1] = (T[:, 1] & 0x0F) | 0x50
    if kind == 'heavy':
        T[rng.rand(n) < 0.3] = T[5]
    if kind == 'steps':
        T[:, 0] = 0; T[:, 1] = rng.randint(1, 5, size=n); T[:, 2] = 4; T[:, 3] = rng.randint(0x4D, 0x8D, size=n)
    ops.sort_config(ctx, level_bits=bits)
    try:
        before = ops.sort_counters(ctx)
        d_T = _dev(ctx, T.ravel())
        perm = ctx.to_numpy(ops.argsort_rows(ctx, d_T, n, C), np.uint32).astype(np.int64)
        after = ops.sort_counters(ctx)
        want = O.argsort_rows(T)
        assert np.array_equal(perm, want)
        if kind == 'heavy': assert after[1] == before[1] + 1, 'a value heavier than a chunk must go to the LSD passes'
        elif kind != 'steps': assert after[0] == before[0] + 1, 'the MSD partition did not run'
        perm2, key, skey, uniq, nu = ops.unique_rows(ctx, d_T, n, C)
        assert np.array_equal(ctx.to_numpy(perm2, np.uint32).astype(np.int64), want)
        S = T[want]
        head = np.ones(n, dtype=bool); head[1:] = (S[1:] != S[:-1]).any(axis=1)
        gid = np.cumsum(head) - 1
        assert nu == int(head.sum())
        assert np.array_equal(ctx.to_numpy(skey, np.uint32).astype(np.int64), gid)
        rkey = np.empty(n, dtype=np.int64); rkey[want] = gid
        assert np.array_equal(ctx.to_numpy(key, np.uint32).astype(np.int64), rkey)
        assert np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), S[head])
    finally:
        ops.sort_config(ctx)


def _rows_in_prefix_groups(rng, n, C, sizes):
    """Rows that tie on their first 6 bytes in groups of the given sizes (the 32-bit round 0 of the sort cannot separate them),
    members differing anywhere behind -- the last byte, the middle, right behind the prefix -- or not at all; shuffled."""
    rows = []
    while len(rows) < n:
        k = int(rng.choice(sizes))
        head = rng.randint(0, 256, size=min(6, C - 1)).astype(np.uint8)
        distinct = max(1, int(rng.randint(1, k + 1)))
        tails = rng.randint(0, 256, size=(distinct, C - len(head))).astype(np.uint8)
        if distinct > 1: tails[1] = tails[0]; tails[1, -1] ^= 1                      # differ in the last byte only
        if distinct > 2: tails[2] = tails[0]; tails[2, 0] ^= 0x80                    # ... right behind the prefix
        for t in tails[rng.randint(0, distinct, size=k)]:
            rows.append(np.concatenate([head, t]))
    T = np.array(rows[:n], dtype=np.uint8)
    return T[rng.permutation(n)]


@pytest.mark.parametrize('form', ['lsd', 'msd'])
@pytest.mark.parametrize('C', [9, 13, 38, 113])
@pytest.mark.parametrize('sizes', [(1, 2), (2, 3, 5, 31, 32), (33, 40, 2), (200,)], ids=['pairs', 'to-32', 'beyond-32', 'long'])
def test_sort_groups_that_tie_on_the_prefix(ctx, C, sizes, form, request):
    """Enough rows for the 32-bit round 0 (>= 65536): groups of up to 32 rows are settled a lane per group by whole rows
    (segment_sort_kernel), longer ones by the radix refinement rounds, mixed in one table; order, duplicate flags (through unique /
    inverse) and stability against numpy."""
    n = 70_000
    rng = np.random.RandomState(C * 7 + len(sizes))
    T = _rows_in_prefix_groups(rng, n, C, sizes)
    d_T = _dev(ctx, T.ravel())
    ops.sort_config(ctx, msd_min_rows=1 if form == 'msd' else -1)
    request.addfinalizer(lambda: ops.sort_config(ctx))
    perm = ctx.to_numpy(ops.argsort_rows(ctx, d_T, n, C), np.uint32).astype(np.int64)
    assert np.array_equal(perm, O.argsort_rows(T))
    perm2, key, skey, uniq, nu = ops.unique_rows(ctx, d_T, n, C)
    ru, rkey = O.unique_rows(T)
    assert nu == len(ru) and np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), ru)
    assert np.array_equal(ctx.to_numpy(key, np.uint32).astype(np.int64), rkey)
    assert np.array_equal(ctx.to_numpy(perm2, np.uint32).astype(np.int64), np.argsort(rkey, kind='stable'))


@pytest.mark.parametrize('n,C', [(1, 1), (100, 38), (5000, 113), (3000, 227), (4097, 25), (10, 300), (50000, 14)])
@pytest.mark.parametrize('isz', [1, 2, 4, 8])
def test_gather_rows(ctx, n, C, isz):
    rng = np.random.RandomState(n + C + isz)
    rows = min(n, 200) if isz == 1 else n
    T = rng.randint(0, 256, size=(rows, C)).astype(np.uint8)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[isz]
    idx = rng.randint(0, min(rows, np.iinfo(dt).max + 1), size=n + 13).astype(dt)
    out = ops.gather_rows(ctx, _dev(ctx, T.ravel()), rows, C, _dev(ctx, idx))
    assert np.array_equal(ctx.to_numpy(out).reshape(-1, C), T[idx.astype(np.int64)])


@pytest.mark.parametrize('dt', [np.uint8, np.uint16, np.uint32, np.uint64])
def test_gather_items(ctx, dt):
    rng = np.random.RandomState(5)
    src = rng.randint(0, np.iinfo(dt).max, size=7001, dtype=np.uint64).astype(dt)
    idx = rng.randint(0, 7001, size=9000).astype(np.uint32)
    out = ops.gather_rows(ctx, _dev(ctx, src).view(ctx.torch.uint8), 7001, src.dtype.itemsize, _dev(ctx, idx))
    assert np.array_equal(ctx.to_numpy(out).view(dt), src[idx])


def test_qname_column_stack_sort_unique(ctx):
    rng = np.random.RandomState(11)
    n = 6000
    cols = [rng.randint(0, 4, n).astype(np.uint8), rng.randint(0, 300, n).astype(np.uint16),
            rng.randint(0, 70000, n).astype(np.uint32), rng.randint(0, 3, n).astype(np.uint8)]
    cols[2][::7] = cols[2][0]
    d_cols = [_dev(ctx, c) for c in cols]
    common = 4
    rows = ops.stack_columns(ctx, d_cols, common)
    stacked = np.dstack(cols)[0]
    be = stacked.astype('>u4').view(np.uint8).reshape(n, -1)
    assert np.array_equal(ctx.to_numpy(rows).reshape(n, -1), be)
    # lexicographic field order == memcmp order of the big-endian rows
    perm = ops.argsort_rows(ctx, rows, n, 4 * common)
    ref = np.lexsort([stacked[:, c] for c in range(3, -1, -1)])
    assert np.array_equal(ctx.to_numpy(perm, np.uint32).astype(np.int64), ref)
    perm, key, skey, uniq, nu = ops.unique_rows(ctx, rows, n, 4 * common)
    ru, rinv = np.unique(stacked, axis=0, return_inverse=True)
    assert nu == len(ru) and np.array_equal(ctx.to_numpy(key, np.uint32), rinv.ravel())
    for c in range(4):
        col = ops.unstack_column(ctx, uniq, nu, 4, common, c, cols[c].dtype.itemsize)
        assert np.array_equal(ctx.to_numpy(col, cols[c].dtype), ru[:, c].astype(cols[c].dtype))


def test_lower_bound_rows_and_single_rank_global_sort(ctx):
    from uq_amd import dist as uqdist
    rng = np.random.RandomState(21)
    n, C = 20000, 38
    T = _rows_with_dups(rng, n, C, 3000)
    order = O.argsort_rows(T)
    S = T[order]
    probes = np.concatenate([S[::997], rng.randint(0, 256, size=(40, C)).astype(np.uint8), np.zeros((1, C), np.uint8), np.full((1, C), 255, np.uint8)])
    pos = ops.lower_bound_rows(ctx, _dev(ctx, S.ravel()), n, C, _dev(ctx, probes.ravel()), len(probes))
    keys = [bytes(r) for r in S]
    import bisect
    assert ctx.to_numpy(pos).tolist() == [bisect.bisect_left(keys, bytes(p)) for p in probes]
    res = uqdist.global_sort_rows(uqdist.HipRows(ctx), _dev(ctx, T.ravel()), n, C, 1000)
    assert res['rows'] == n and res['offset'] == 0
    assert np.array_equal(ctx.to_numpy(res['gidx']), order + 1000)
    assert np.array_equal(ctx.to_numpy(res['table']).reshape(n, C), S)
    g = uqdist.dist_gather_rows(uqdist.HipRows(ctx), _dev(ctx, T.ravel()), n, C, [1000, 1000 + n], res['gidx'])
    assert np.array_equal(ctx.to_numpy(g).reshape(n, C), S)
    back = uqdist.dist_scatter_rows(uqdist.HipRows(ctx), res['table'], C, [1000, 1000 + n], res['gidx'])
    assert np.array_equal(ctx.to_numpy(back).reshape(n, C), T)


@pytest.mark.parametrize('n,ndest', [(1, 1), (5, 3), (4096, 8), (4097, 8), (100_003, 2), (70_000, 16), (300_000, 5), (12_289, 1)])
def test_partition_order(ctx, n, ndest):
    """uq_partition_order (what goes in front of every all-to-all of uq_amd.dist): the stable partition of positions by a one-byte
    destination == numpy's stable argsort of the bytes, counts == bincount; skewed and missing destinations."""
    rng = np.random.RandomState(n + ndest)
    for kind in ('uniform', 'skewed', 'one'):
        if kind == 'uniform': d = rng.randint(0, ndest, size=n)
        elif kind == 'skewed': d = np.minimum(rng.geometric(0.6, size=n) - 1, ndest - 1)
        else: d = np.full(n, ndest - 1)
        d = d.astype(np.uint8)
        order, counts = ops.partition_order(ctx, _dev(ctx, d), n, ndest)
        assert np.array_equal(ctx.to_numpy(order, np.uint32).astype(np.int64), np.argsort(d, kind='stable'))
        assert np.array_equal(ctx.to_numpy(counts.view(ctx.torch.uint8), np.int64), np.bincount(d, minlength=ndest))
    from uq_amd._lib import UqHipError
    with pytest.raises(UqHipError):
        ops.partition_order(ctx, _dev(ctx, np.zeros(4, dtype=np.uint8)), 4, 17)


@pytest.mark.parametrize('n,C,nd,prefix', SORT_CASES, ids=lambda v: str(v))
def test_unique_rows_of_groups(ctx, n, C, nd, prefix):
    """The distinct rows of a sorted table from the group ids its sort left (uq_unique_rows_of_groups) == numpy.unique's table."""
    rng = np.random.RandomState(n * 11 + C)
    T = _rows_with_dups(rng, n, C, nd, prefix)
    d_T = _dev(ctx, T.ravel())
    perm, _, skey, _, nu = ops.unique_rows(ctx, d_T, n, C, want_key=False, want_unique=False)
    S = ops.gather_rows(ctx, d_T, n, C, perm)
    uniq = ops.unique_rows_of_groups(ctx, S, n, C, skey, nu)
    ru, _ = O.unique_rows(T)
    assert nu == len(ru) and np.array_equal(ctx.to_numpy(uniq).reshape(nu, C), ru)
    assert ctx.torch.equal(ops.unique_rows_of_groups(ctx, d_T, n, C, skey, nu, perm=perm), uniq)      # ... without moving the table first


@pytest.mark.parametrize('C', [1, 3, 4, 7, 8, 13, 16, 38, 64, 113, 200])
@pytest.mark.parametrize('itemsize', [4, 8])
def test_scatter_rows(ctx, C, itemsize):
    """uq_scatter_rows: out[index[j] - base] = values[j] for every row width class (1 / 2 / 4 / 16 / 64 lanes per row) and both index
    widths; an index outside the output is reported, not written."""
    be = __import__('uq_amd.dist', fromlist=['HipRows']).HipRows(ctx)
    rng = np.random.RandomState(C * 2 + itemsize)
    n, base = 5003, 77
    V = rng.randint(0, 256, size=(n, C)).astype(np.uint8)
    perm = rng.permutation(n)
    idx = (perm + base).astype(np.uint32 if itemsize == 4 else np.int64)
    out = be.scatter_rows(_dev(ctx, V.ravel()), n, C, _dev(ctx, idx.view(np.uint8)).view(ctx.torch.int32 if itemsize == 4 else ctx.torch.int64), base, n)
    want = np.empty_like(V)
    want[perm] = V
    assert np.array_equal(ctx.to_numpy(out).reshape(n, C), want)
    idx[1234] = base + n
    idx[4000] = base - 1 if itemsize == 8 else base + n + 5
    with pytest.raises(RuntimeError, match='entry 1234'):
        be.scatter_rows(_dev(ctx, V.ravel()), n, C, _dev(ctx, idx.view(np.uint8)).view(ctx.torch.int32 if itemsize == 4 else ctx.torch.int64), base, n)
    assert len(be.scatter_rows(_dev(ctx, V[:0].ravel()), 0, C, _dev(ctx, idx[:1].view(np.uint8)).view(ctx.torch.int32 if itemsize == 4 else ctx.torch.int64)[:0], base, 0)) == 0


UNPACK_CASES = [('fixed', 2000, 100, {}, {}), ('var_ntrick', 3000, (36, 301), dict(n_rate=1), {}),
                ('var_notricks', 2000, (20, 150), dict(n_rate=2), dict(notricks=True)),
                ('var_pad', 500, (1, 40), dict(n_rate=2), dict(notricks=True, pad=True)), ('len1', 50, 1, {}, {})]


@pytest.mark.parametrize('name,n,length,kw,dk', UNPACK_CASES, ids=[c[0] for c in UNPACK_CASES])
def test_unpack_roundtrip(ctx, name, n, length, kw, dk):
    from uq_amd import synth
    spec = synth.Spec(20261010, length, **kw)
    host = synth.fastq_array(spec, n)
    hls = oracle_c.index_lines(host)
    st = oracle_c.stats(host, hls, 0, n)
    d = O.decide(O.histogram_to_static_qualities(st['counts'], st['first_seen']), st['len_min'], st['len_max'], **dk)
    rd, rq, _ = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                              d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'],
               bits_per_quality=d['bits_per_quality'], variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'])
    p = ops.make_unpack_params(cfg)
    seq, qt, ln, bad = ops.unpack(ctx, _dev(ctx, rd.ravel()), _dev(ctx, rq.ravel()), n, p)
    assert ops.bad_index(bad) is None
    rs, rqt, rln, rbad = oracle_c.unpack(rd, rq, cfg)
    assert rbad is None
    assert np.array_equal(ctx.to_numpy(ln, np.uint32), rln)
    assert np.array_equal(ctx.to_numpy(seq).reshape(n, -1), rs)
    assert np.array_equal(ctx.to_numpy(qt).reshape(n, -1), rqt)
    # and the text really is the input
    lines = host.tobytes().split(b'\n')
    S = ctx.to_numpy(seq).reshape(n, -1); Q = ctx.to_numpy(qt).reshape(n, -1); L = ctx.to_numpy(ln, np.uint32)
    for r in range(0, n, max(1, n // 97)):
        assert S[r, :L[r]].tobytes() == lines[4 * r + 1]
        assert Q[r, :L[r]].tobytes() == lines[4 * r + 3]


DECODE_CASES = UNPACK_CASES + [('long_rows', 40, (2000, 9000), dict(n_rate=1), {}), ('ragged', 1031, (30, 60), {}, {}),
                               ('one_tile_budget', 300, 508, dict(n_rate=3), dict(notricks=True))]


@pytest.mark.parametrize('name,n,length,kw,dk', DECODE_CASES, ids=[c[0] for c in DECODE_CASES])
def test_decode_fastq_is_the_input_and_the_two_pass_text(ctx, name, n, length, kw, dk):
    """uq_decode_fastq (rows -> text in one kernel) against uq_unpack + uq_emit_fastq and against the FASTQ the rows
    were packed from (oracle pack, host QNAME analysis)."""
    from uq_amd import qname, synth
    if name == 'long_rows':
        rng = np.random.default_rng(5)
        recs = []
        for i in range(n):
            L = int(rng.integers(length[0], length[1]))
            recs.append(b'@r:%d:%d\n' % (1000 - i, i % 3) + bytes(rng.choice(list(b'ACGTN'), L, p=[.24, .25, .25, .25, .01]).astype(np.uint8)) + b'\n+\n' +
                        bytes(rng.integers(35, 75, L).astype(np.uint8)) + b'\n')
        host = np.frombuffer(b''.join(recs), dtype=np.uint8)
    else:
        host = synth.fastq_array(synth.Spec(20261011, length, **kw), n)
    hls = oracle_c.index_lines(host)
    st = oracle_c.stats(host, hls, 0, n)
    d = O.decide(O.histogram_to_static_qualities(st['counts'], st['first_seen']), st['len_min'], st['len_max'], **dk)
    rd, rq, _ = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                              d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    prefix, suffix, separators, columns, arrays = qname.analyse(qname.qname_lines(host, hls, n))
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'],
               bits_per_quality=d['bits_per_quality'], variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'],
               QNAME_prefix=prefix, QNAME_suffix=suffix, QNAME_separators=separators, QNAME_columns=columns)
    cols = [_dev(ctx, np.ascontiguousarray(a)) for a in arrays]
    dna, qual = _dev(ctx, rd.ravel()), _dev(ctx, rq.ravel())
    text, bad = ops.decode_fastq(ctx, cfg, cols, dna, qual, n)
    assert bad is None
    assert ctx.to_numpy(text).tobytes() == host.tobytes()
    seq, qt, ln, ubad = ops.unpack(ctx, dna, qual, n, ops.make_unpack_params(cfg))
    two = ops.emit_fastq(ctx, cfg, cols, seq, qt, ln, n)
    assert ctx.to_numpy(two).tobytes() == host.tobytes()
    if d['variable_read_lengths'] and n > 10:
        # a row without its sentinel: both paths name it
        rd2 = rd.copy(); rd2[7, :] = 0
        text, bad = ops.decode_fastq(ctx, cfg, cols, _dev(ctx, rd2.ravel()), qual, n)
        assert text is None and bad == 7


@pytest.mark.parametrize('L', [8, 40, 65, 66, 73, 100, 150, 152, 250, 303, 700], ids=lambda v: 'L%d' % v)
@pytest.mark.parametrize('with_n', [False, True], ids=['acgt', 'n-trick'])
def test_decode_fixed_lengths_chunks_per_lane(ctx, L, with_n):
    """Fixed-length tables on both sides of the length where the tile kernel's lanes take five 8-character chunks of a line instead of one
    (66 characters): the text is the input; then the same tables with random bytes for quality rows -- codes beyond the alphabet, which no encoder
    writes -- decode to the same text through both decoders (such a code becomes byte 0, as through the tables)."""
    from uq_amd import qname
    rng = np.random.default_rng(L * 2 + with_n)
    n = 1500
    recs = []
    for i in range(n):
        seq = rng.choice(np.frombuffer(b'ACGT', np.uint8), L)
        q = rng.integers(38, 38 + 41, L).astype(np.uint8)
        if with_n:
            at = rng.random(L) < 0.03
            seq[at] = ord('N'); q[at] = 35
        recs.append(b'@m%d:%d:%d\n' % (i % 7, i, int(rng.integers(0, 10 ** int(rng.integers(1, 9))))) + bytes(seq) + b'\n+\n' + bytes(q) + b'\n')
    host = np.frombuffer(b''.join(recs), dtype=np.uint8)
    hls = oracle_c.index_lines(host)
    st = oracle_c.stats(host, hls, 0, n)
    d = O.decide(O.histogram_to_static_qualities(st['counts'], st['first_seen']), st['len_min'], st['len_max'])
    assert not d['variable_read_lengths'] and d['bits_per_quality'] == 6 and bool(d['N_qual']) == with_n
    rd, rq, _ = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                              d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    prefix, suffix, separators, columns, arrays = qname.analyse(qname.qname_lines(host, hls, n))
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'],
               bits_per_quality=d['bits_per_quality'], variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'],
               QNAME_prefix=prefix, QNAME_suffix=suffix, QNAME_separators=separators, QNAME_columns=columns)
    cols = [_dev(ctx, np.ascontiguousarray(a)) for a in arrays]
    ops.scribble_lds(ctx, 0xA5A5A5A5)
    text, bad = ops.decode_fastq(ctx, cfg, cols, _dev(ctx, rd.ravel()), _dev(ctx, rq.ravel()), n)
    assert bad is None
    assert ctx.to_numpy(text).tobytes() == host.tobytes()
    rq2 = rq.copy()
    rows = rng.random(n) < 0.3
    rq2[rows] = rng.integers(0, 256, size=(int(rows.sum()), rq.shape[1])).astype(np.uint8)
    dq2 = _dev(ctx, rq2.ravel())
    text2, bad = ops.decode_fastq(ctx, cfg, cols, _dev(ctx, rd.ravel()), dq2, n)
    seq, qt, ln, ubad = ops.unpack(ctx, _dev(ctx, rd.ravel()), dq2, n, ops.make_unpack_params(cfg))
    two = ops.emit_fastq(ctx, cfg, cols, seq, qt, ln, n)
    a, b = ctx.to_numpy(text2), ctx.to_numpy(two)
    assert a.tobytes() == b.tobytes()
    assert (a == 0).any()                      # 23 of the 64 codes are beyond the alphabet
    keep = ~np.repeat(rows, 1)                 # the rows left alone still decode to the input
    offs = hls[::4]
    for r in np.flatnonzero(keep)[:200]:
        assert a[offs[r]:offs[r + 1]].tobytes() == host[offs[r]:offs[r + 1]].tobytes()


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('UQ_DECODE_FUZZ_N', '40'))))       # UQ_DECODE_FUZZ_N=1000 for a longer hunt
def test_decode_fixed_lengths_random(ctx, seed):
    """Random fixed-length tables for the tile kernel's multi-chunk instances (the CLI fuzz keeps fixed lengths below 40): lengths 40 - 600 on
    both sides of the switch at 66, 3 - 64 qualities (2 - 6 bits), 2- and 3-bit base alphabets, with and without the N-trick, one to several
    tiles with a ragged last one, QNAME columns of one to four fields -- decode(pack(x)) == x through uq_decode_fastq and the two-pass decoder."""
    from uq_amd import qname
    rng = np.random.default_rng(424_200 + seed)
    L = int(rng.integers(40, 600))
    n = int(rng.integers(1, max(2, 120_000 // L)))
    nq = int(rng.choice([3, 4, 7, 8, 15, 16, 20, 31, 41, 64]))
    six = rng.random() < 0.25
    with_n = (not six) and rng.random() < 0.5
    nf = int(rng.integers(1, 5))
    B = np.frombuffer(b'ACGTRY' if six else b'ACGT', np.uint8)
    recs = []
    for i in range(n):
        seq = rng.choice(B, L)
        q = rng.integers(40, 40 + nq, L).astype(np.uint8)
        if with_n:
            at = rng.random(L) < 0.03
            seq[at] = ord('N'); q[at] = 35
        name = b'@r' + b':'.join(b'%d' % int(rng.integers(0, 10 ** int(rng.integers(1, 9)))) for _ in range(nf)) + b':%d' % i
        recs.append(name + b'\n' + bytes(seq) + b'\n+\n' + bytes(q) + b'\n')
    host = np.frombuffer(b''.join(recs), dtype=np.uint8)
    hls = oracle_c.index_lines(host)
    st = oracle_c.stats(host, hls, 0, n)
    d = O.decide(O.histogram_to_static_qualities(st['counts'], st['first_seen']), st['len_min'], st['len_max'])
    if d['N_qual'] and max(d['N_qual'].values()) >= len(d['qualities']): return          # Q9: not decodable by the reference either
    rd, rq, _ = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                              d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    try:
        prefix, suffix, separators, columns, arrays = qname.analyse(qname.qname_lines(host, hls, n))
    except qname.QnameError:
        return
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'],
               bits_per_quality=d['bits_per_quality'], variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'],
               QNAME_prefix=prefix, QNAME_suffix=suffix, QNAME_separators=separators, QNAME_columns=columns)
    cols = [_dev(ctx, np.ascontiguousarray(a)) for a in arrays]
    ops.scribble_lds(ctx, 0x3C3C3C3C ^ seed)
    dna, qual = _dev(ctx, rd.ravel()), _dev(ctx, rq.ravel())
    text, bad = ops.decode_fastq(ctx, cfg, cols, dna, qual, n)
    assert bad is None and ctx.to_numpy(text).tobytes() == host.tobytes(), (L, n, nq, six, with_n)
    seq, qt, ln, ubad = ops.unpack(ctx, dna, qual, n, ops.make_unpack_params(cfg))
    assert ctx.to_numpy(ops.emit_fastq(ctx, cfg, cols, seq, qt, ln, n)).tobytes() == host.tobytes()


@pytest.mark.parametrize('variable', [False, True], ids=['fixed', 'variable'])
@pytest.mark.parametrize('with_n', [False, True, 'six'], ids=['acgt', 'n-trick', 'six-bases'])
@pytest.mark.parametrize('nq', [3, 4, 6, 12, 20, 41, 70], ids=lambda v: 'q%d' % v)
def test_decode_quality_widths(ctx, nq, with_n, variable):
    """Tables with the lookup-free alphabet at every quality width (2 .. 7 bits): uq_decode_fastq runs the instance of the tile kernel
    (fixed lengths) or of the streaming kernel (variable lengths) compiled for that width and N-trick (2 .. 6 bits) or the run-time
    one (7); several tiles, a ragged last one.  'six-bases': a 3-bit base alphabet (ACGTRY), lookup-free too (v_perm over eight characters)."""
    from uq_amd import qname
    rng = np.random.default_rng(nq * 2 + (with_n is True))
    n = 3001
    recs = []
    for i in range(n):
        L = int(rng.integers(1, 161)) if variable else 151
        seq = rng.choice(np.frombuffer(b'ACGTRY' if with_n == 'six' else b'ACGT', np.uint8), L)
        q = rng.integers(40, 40 + nq, L).astype(np.uint8)
        if with_n is True:
            at = rng.random(L) < 0.02
            seq[at] = ord('N'); q[at] = 35
        recs.append(b'@x%d:%d/1\n' % (i, int(rng.integers(0, 99999))) + bytes(seq) + b'\n+\n' + bytes(q) + b'\n')
    host = np.frombuffer(b''.join(recs), dtype=np.uint8)
    hls = oracle_c.index_lines(host)
    st = oracle_c.stats(host, hls, 0, n)
    d = O.decide(O.histogram_to_static_qualities(st['counts'], st['first_seen']), st['len_min'], st['len_max'])
    assert d['bits_per_base'] == (3 if with_n == 'six' else 2) and bool(d['variable_read_lengths']) == variable and bool(d['N_qual']) == (with_n is True)
    rd, rq, _ = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                              d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    prefix, suffix, separators, columns, arrays = qname.analyse(qname.qname_lines(host, hls, n))
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'],
               bits_per_quality=d['bits_per_quality'], variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'],
               QNAME_prefix=prefix, QNAME_suffix=suffix, QNAME_separators=separators, QNAME_columns=columns)
    cols = [_dev(ctx, np.ascontiguousarray(a)) for a in arrays]
    ops.scribble_lds(ctx, 0x5A5A5A5A)
    text, bad = ops.decode_fastq(ctx, cfg, cols, _dev(ctx, rd.ravel()), _dev(ctx, rq.ravel()), n)
    assert bad is None
    assert ctx.to_numpy(text).tobytes() == host.tobytes()
