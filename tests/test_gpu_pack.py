"""GPU parity: record index, pass-1 statistics, packers and the synthetic generator, through the C ABI,
against the oracle on the same seeded inputs.  Bit-exact (byte / integer work)."""
import json
import os

import numpy as np
import pytest

import oracle_c
import uq_oracle as O
from uq_amd import ops, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
S = 20261003


def _index(ctx, d_buf):
    nlines = ops.count_lines(ctx, d_buf)
    ls = ops.index_lines(ctx, d_buf, nlines)
    return nlines, ls


def _decide_from_stats(hs, notricks=False, pad=False, first_seen=None):
    sq = O.histogram_to_static_qualities(hs.counts, first_seen)
    return O.decide(sq, hs.len_min, hs.len_max, notricks, pad)


def _gpu_pack(ctx, d_buf, ls, n, d, max_record_bytes):
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                             d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'],
                             d['dna_max'], max_record_bytes)
    dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
    return (ctx.to_numpy(dna).reshape(n, -1), ctx.to_numpy(qual).reshape(n, -1), ops.bad_index(bad))


@pytest.mark.parametrize('n,length,kw', [
    (1, 1, {}), (3, 7, {}), (1000, 100, {}), (5000, 150, {}), (777, (36, 301), dict(n_rate=1)),
    (300, 50, dict(n_rate=3, n_qual_exclusive=False)), (513, (1, 9), {}),
])
def test_synth_matches_host(ctx, n, length, kw):
    spec = synth.Spec(S + 2, length, **kw)
    host = synth.fastq_array(spec, n, first=5)
    dev = ops.synth_fastq(ctx, spec, 5, n)
    assert np.array_equal(ctx.to_numpy(dev), host)


@pytest.mark.parametrize('misalign', [0, 1, 7, 15])
def test_index_lines(ctx, misalign):
    spec = synth.Spec(S + 3, (36, 120), n_rate=1)
    host = synth.fastq_array(spec, 20000)
    t = ctx.torch
    backing = ctx.empty(host.size + 64)
    d_buf = backing[misalign:misalign + host.size]
    d_buf.copy_(t.from_numpy(host))
    nlines, ls = _index(ctx, d_buf)
    ref = oracle_c.index_lines(host)
    assert nlines == len(ref) - 1 == 80000
    assert np.array_equal(ctx.to_numpy(ls, np.uint64), ref)


def test_index_edge_cases(ctx):
    t = ctx.torch
    for data in [b'', b'\n', b'abc', b'abc\n', b'\n\n\n\n', b'a\nbb\nccc\ndddd', b'x' * 40000 + b'\n' + b'y' * 17]:
        host = np.frombuffer(data, dtype=np.uint8)
        d = ctx.empty(max(len(data), 1))[:len(data)]
        if len(data): d.copy_(t.from_numpy(host.copy()))
        nlines = ops.count_lines(ctx, d)
        assert nlines == data.count(b'\n')
        ls = ops.index_lines(ctx, d, nlines)
        assert np.array_equal(ctx.to_numpy(ls, np.uint64), oracle_c.index_lines(host))


@pytest.mark.parametrize('misalign', [0, 5])
def test_index_dense_newlines(ctx, misalign):
    """The census keeps a tile's newlines in a list slot of 1024 entries; tiles with more (lines shorter than 16 bytes)
    make the index fall back to the bitmap form.  Line lengths around that limit, tiles of nothing but newlines, and a
    mix of dense and sparse tiles in one buffer."""
    t = ctx.torch
    rng = np.random.default_rng(7)
    cases = [b'\n' * 70000, b'ab\n' * 40000, (b'x' * 15 + b'\n') * 5000, (b'x' * 14 + b'\n') * 5000, (b'x' * 16 + b'\n') * 5000,
             (b'x' * 15 + b'\n') * 1024 + b'\n' + (b'y' * 15 + b'\n') * 3000,                      # exactly 1024, then 1025 in a tile
             b'q' * 50000 + b'\n' * 3000 + b'r' * 40001 + b'\n',
             b''.join(b'z' * int(k) + b'\n' for k in rng.integers(0, 40, 30000))]
    for data in cases:
        host = np.frombuffer(data, dtype=np.uint8)
        backing = ctx.empty(host.size + 64)
        d = backing[misalign:misalign + host.size]
        d.copy_(t.from_numpy(host.copy()))
        ref = oracle_c.index_lines(host)
        nlines = ops.count_lines(ctx, d)
        assert nlines == len(ref) - 1
        assert np.array_equal(ctx.to_numpy(ops.index_lines(ctx, d, nlines), np.uint64), ref)
        # and without the census of this buffer in the context (another buffer counted in between)
        ops.count_lines(ctx, backing[:17])
        assert np.array_equal(ctx.to_numpy(ops.index_lines(ctx, d, nlines), np.uint64), ref)


@pytest.mark.parametrize('n,length,kw', [
    (2000, 100, {}), (3000, (36, 301), dict(n_rate=1)), (500, 50, dict(n_rate=3, n_qual_exclusive=False)),
    (1, 5, {}), (2, (1, 3), {}),
])
def test_stats(ctx, n, length, kw):
    spec = synth.Spec(S + 4, length, **kw)
    host = synth.fastq_array(spec, n)
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, n)
    hs = ops.stats_fetch(ctx, st)
    ref = oracle_c.stats(host, oracle_c.index_lines(host), 0, n)
    assert np.array_equal(hs.counts, ref['counts'])
    assert (hs.len_min, hs.len_max, hs.max_record_bytes) == (ref['len_min'], ref['len_max'], ref['max_record_bytes'])
    assert hs.bad_plus is None and hs.bad_len is None
    fs = ops.first_occurrence(ctx, d_buf, ls, 0, n)
    assert np.array_equal(fs, ref['first_seen'])


def test_stats_unusual_bytes_and_bad_records(ctx):
    recs = [b'@r:1:1\nACGTZ\x80\xff\n+\n!~\x01\x02\xfe\xff\x7f\n', b'@r:2:2\nAC\n-\nII\n', b'@r:3:3\nACG\n+\nII\n', b'@r:4:4\nA\n+\nI\n']
    data = b''.join(recs)
    host = np.frombuffer(data, dtype=np.uint8).copy()
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, 4)
    hs = ops.stats_fetch(ctx, st)
    ref = oracle_c.stats(host, oracle_c.index_lines(host), 0, 4)
    assert np.array_equal(hs.counts, ref['counts'])
    assert hs.bad_plus == 1 and hs.bad_len == 2
    # shards accumulate into the same struct
    st2 = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st2, d_buf, ls, 0, 1)
    ops.stats_accumulate(ctx, st2, d_buf, ls, 1, 3)
    hs2 = ops.stats_fetch(ctx, st2)
    assert np.array_equal(hs2.counts, ref['counts']) and hs2.bad_plus == 1 and hs2.bad_len == 2


@pytest.mark.parametrize('case', ['lower', 'mixed', 'phred64', 'iupac'])
def test_stats_window_hints(ctx, case):
    """The LDS count tables are windows placed from the first record; every byte must still be counted exactly."""
    rng = np.random.default_rng(11)
    alpha = {'lower': b'acgtn', 'mixed': b'ACGTacgtN', 'phred64': b'ACGT', 'iupac': b'ACGTNRYKMSWBDHV'}[case]
    qlo, qhi = (64, 105) if case == 'phred64' else (33, 74)
    recs = []
    for i in range(3000):
        L = int(rng.integers(1, 120))
        recs.append(b'@r:%d\n' % i + bytes(rng.choice(np.frombuffer(alpha, np.uint8), L)) + b'\n+\n' + bytes(rng.integers(qlo, qhi, L, dtype=np.uint8)) + b'\n')
    host = np.frombuffer(b''.join(recs), dtype=np.uint8).copy()
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, 3000)
    hs = ops.stats_fetch(ctx, st)
    ref = oracle_c.stats(host, oracle_c.index_lines(host), 0, 3000)
    assert np.array_equal(hs.counts, ref['counts'])
    assert (hs.len_min, hs.len_max) == (ref['len_min'], ref['len_max'])


def _check_index_stats(ctx, host, misalign=0):
    t = ctx.torch
    backing = ctx.empty(host.size + 64)
    d_buf = backing[misalign:misalign + host.size]
    d_buf.copy_(t.from_numpy(host))
    nlines = ops.count_lines(ctx, d_buf)
    hls = oracle_c.index_lines(host)
    assert nlines == len(hls) - 1
    ls, st = ops.index_and_stats(ctx, d_buf, nlines)
    assert np.array_equal(ctx.to_numpy(ls, np.uint64), hls)
    hs = ops.stats_fetch(ctx, st)
    n = nlines // 4
    ref = oracle_c.stats(host, hls, 0, n)
    assert np.array_equal(hs.counts, ref['counts'])
    assert (hs.len_min, hs.len_max, hs.max_record_bytes) == (ref['len_min'], ref['len_max'], ref['max_record_bytes'])
    assert (hs.bad_plus, hs.bad_len) == (ref['bad_plus'], ref['bad_len'])


@pytest.mark.parametrize('n,length,kw,mis', [
    (1, 5, {}, 0), (3, 7, {}, 3), (20000, 100, {}, 0), (30000, 150, {}, 9), (9000, (36, 301), dict(n_rate=1), 15),
    (4000, (1, 9), {}, 1), (700, 50, dict(n_rate=3, n_qual_exclusive=False), 0),
])
def test_index_stats_against_oracle(ctx, n, length, kw, mis):
    _check_index_stats(ctx, synth.fastq_array(synth.Spec(S + 7, length, **kw), n), mis)


def test_index_stats_hard_inputs(ctx):
    rng = np.random.RandomState(3)
    # reads far longer than the 4 KiB halo (counted from HBM), mixed with short ones
    recs = []
    for i in range(60):
        L = int(rng.choice([5, 3000, 9000, 40, 17000]))
        seq = ''.join('ACGTN'[k] for k in rng.randint(0, 5, L)); q = ''.join(chr(33 + k) for k in rng.randint(0, 41, L))
        recs.append('@long:%d:%d\n%s\n+\n%s\n' % (i % 3, i, seq, q))
    _check_index_stats(ctx, np.frombuffer(''.join(recs).encode(), dtype=np.uint8).copy(), 5)
    # unusual bytes (outside both LDS windows), a bad '+' line and a length mismatch
    recs = [b'@r:1:1\nACGTZ\x80\xff\n+\n!~\x01\x02\xfe\xff\x7f\n', b'@r:2:2\nAC\n-\nII\n', b'@r:3:3\nACG\n+\nII\n', b'@r:4:4\nA\n+\nI\n'] * 300
    _check_index_stats(ctx, np.frombuffer(b''.join(recs), dtype=np.uint8).copy())
    # thousands of one-byte lines per tile: the newline lists overflow, the bitmap form of the index runs
    _check_index_stats(ctx, np.frombuffer(b'@\nA\n+\nI\n' * 20000, dtype=np.uint8).copy(), 2)
    # Phred+64 style qualities (window moves) and lower-case bases
    recs = [('@p:%d\n%s\n+\n%s\n' % (i, 'acgtn' * 8, ''.join(chr(64 + (i + k) % 41) for k in range(40)))).encode() for i in range(3000)]
    _check_index_stats(ctx, np.frombuffer(b''.join(recs), dtype=np.uint8).copy(), 11)


PACK_CASES = [
    ('fixed100', 3000, 100, {}, {}),
    ('fixed150', 4096 + 37, 150, {}, {}),
    ('len1', 100, 1, {}, {}),
    ('len4_bits8', 64, 4, {}, {}),                          # b*L % 8 == 0, fixed
    ('var_ntrick', 3000, (36, 301), dict(n_rate=1), {}),    # 2-bit DNA via the N-trick, variable, all L mod 4
    ('var_notricks', 3000, (36, 301), dict(n_rate=1), dict(notricks=True)),   # 3-bit ACGNT
    ('var_pad', 1000, (20, 77), dict(n_rate=2), dict(notricks=True, pad=True)),  # 4-bit / 8-bit
    ('n_newcode', 1500, 50, dict(n_rate=3, n_qual_exclusive=False), {}),      # N_qual = 42 (skips a code), 6 bits
    ('short_var', 2000, (1, 12), {}, {}),
]


@pytest.mark.parametrize('name,n,length,kw,dk', PACK_CASES, ids=[c[0] for c in PACK_CASES])
def test_pack_matches_oracle(ctx, name, n, length, kw, dk):
    spec = synth.Spec(S + 5, length, **kw)
    host = synth.fastq_array(spec, n)
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    assert nlines == 4 * n
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, n)
    hs = ops.stats_fetch(ctx, st)
    d = _decide_from_stats(hs, **dk)
    dna, qual, bad = _gpu_pack(ctx, d_buf, ls, n, d, hs.max_record_bytes)
    assert bad is None
    hls = oracle_c.index_lines(host)
    rd, rq, rbad = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'],
                                 d['bits_per_quality'], d['variable_read_lengths'], d['dna_bytes_per_row'],
                                 d['quality_bytes_per_row'])
    assert rbad is None
    assert np.array_equal(dna, rd)
    assert np.array_equal(qual, rq)
    # and the faithful Python loop on a prefix (the C loop is itself checked against it on CPU)
    m = min(n, 200)
    lines = O.read_lines(host.tobytes())
    pd_, pq_ = O.encoder(lines, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'],
                         d['quality_bytes_per_row'], d['bits_per_base'], d['bits_per_quality'],
                         d['variable_read_lengths'], count=m)
    assert np.array_equal(dna[:m], pd_) and np.array_equal(qual[:m], pq_)


def test_pack_q9_carry_path(ctx):
    """3 qualities + one NEW N code -> total_quals = 4 -> 2 bits, N_qual = 4 = 2^b: the `+=` carries (Q9)."""
    recs = []
    rng = np.random.RandomState(7)
    for i in range(400):
        L = 21
        seq = ''.join('ACGTN'[k] for k in rng.choice(5, L, p=[.23, .23, .23, .23, .08]))
        q = ''.join('I' if c == 'N' else '#HI'[rng.randint(3)] for c in seq)
        recs.append('@q:%d:%d\n%s\n+\n%s\n' % (i % 3, i, seq, q))
    data = ''.join(recs).encode()
    host = np.frombuffer(data, dtype=np.uint8).copy()
    lines = O.read_lines(data)
    p1 = O.pass1(lines)
    d = O.decide(p1['static_qualities'], p1['dna_min'], p1['dna_max'])
    assert d['N_qual'] == {'N': 4} and d['bits_per_quality'] == 2
    rd, rq = O.encoder(lines, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'], d['quality_bytes_per_row'],
                       d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'])
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    dna, qual, bad = _gpu_pack(ctx, d_buf, ls, 400, d, 128)
    assert bad is None
    assert np.array_equal(dna, rd) and np.array_equal(qual, rq)


def test_pack_reports_uncoded_symbol(ctx):
    data = b'@a:1\nACGT\n+\nIIII\n@a:2\nACXT\n+\nIIII\n'
    host = np.frombuffer(data, dtype=np.uint8).copy()
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    p = ops.make_pack_params('ACGT', 'I', {}, 2, 2, False, 1, 1, 4, 64)
    dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, 2, p)
    assert ops.bad_index(bad) == 1


@pytest.mark.parametrize('name', ['cfg1_10k_100bp', 'fixed_n_newcode', 'variable_ntrick', 'variable_notricks', 'fixed_pad', 'alpha_iupac_4bit', 'qual_7bit', 'qual_8bit',
                                  'two_ntrick_bases', 'one_base_one_qual', 'alpha_mixed_case', 'var_tiny_alphabets'])
def test_pack_matches_reference_golden(ctx, name):
    """DNA.raw / QUAL.raw written by the reference itself (tests/golden/*.uQ) == the HIP packers' rows."""
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    data = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    cfg, members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
    host = np.frombuffer(data, dtype=np.uint8).copy()
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    n = nlines // 4
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, n)
    hs = ops.stats_fetch(ctx, st)
    fs = ops.first_occurrence(ctx, d_buf, ls, 0, n)
    d = _decide_from_stats(hs, notricks='--notricks' in meta['flags'], pad='--pad' in meta['flags'], first_seen=fs)
    assert d['bases'] == cfg['bases'] and d['qualities'] == cfg['qualities'] and d['N_qual'] == cfg['N_qual']
    assert d['bits_per_base'] == cfg['bits_per_base'] and d['bits_per_quality'] == cfg['bits_per_quality']
    dna, qual, bad = _gpu_pack(ctx, d_buf, ls, n, d, hs.max_record_bytes)
    assert bad is None
    assert np.array_equal(dna, O.unpattern(members['DNA.raw']))
    assert np.array_equal(qual, O.unpattern(members['QUAL.raw']))


@pytest.mark.parametrize('fused', [False, True], ids=['pack', 'pack_stats'])
def test_pack_tiles_sized_from_the_average_record(ctx, fused):
    """Tiles hold R reads with R taken from the AVERAGE record; a tile of unusually long records does not fit the
    stage and is packed in pieces (and, fused, still counted completely)."""
    rng = np.random.default_rng(5)
    recs = []
    for i in range(2600):
        L = 480 if 900 <= i < 1010 or i % 577 == 3 else int(rng.integers(10, 31))      # a run of long reads among short ones
        recs.append(b'@r:%d\n' % i + bytes(rng.choice(np.frombuffer(b'ACGT', np.uint8), L)) + b'\n+\n' +
                    bytes(rng.integers(33, 74, L, dtype=np.uint8)) + b'\n')
    host = np.frombuffer(b''.join(recs), dtype=np.uint8).copy()
    n = len(recs)
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, n)
    hs = ops.stats_fetch(ctx, st)
    d = _decide_from_stats(hs)
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes,
                             avg_record_bytes=host.size // n)
    assert host.size // n < hs.max_record_bytes // 8           # the average really is far below the longest record
    if fused:
        dna, qual, bad, st2 = ops.pack_stats(ctx, d_buf, ls, 0, n, p)
        hs2 = ops.stats_fetch(ctx, st2)
        assert not hs2.incomplete and np.array_equal(hs2.counts, hs.counts) and (hs2.len_min, hs2.len_max) == (hs.len_min, hs.len_max)
    else:
        dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
    assert ops.bad_index(bad) is None
    hls = oracle_c.index_lines(host)
    rd, rq, rbad = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                                 d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    assert np.array_equal(ctx.to_numpy(dna).reshape(n, -1), rd) and np.array_equal(ctx.to_numpy(qual).reshape(n, -1), rq)


FUSED_CASES = [('fixed150', 4096 + 37, 150, {}, {}), ('fixed100', 3000, 100, {}, {}), ('var_ntrick', 5000, (36, 301), dict(n_rate=1), {}),
               ('short_var', 2000, (1, 12), {}, {}),
               # three bits of base (uq.py:479 --notricks keeps N as a fifth base): bins of nine bits, counted pair by pair
               ('var_notricks_ACGNT', 5000, (36, 301), dict(n_rate=1), dict(notricks=True)), ('fixed_notricks_ACGNT', 3000, 100, dict(n_rate=2), dict(notricks=True))]


@pytest.mark.parametrize('name,n,length,kw,dk', FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
def test_pack_stats_fused(ctx, name, n, length, kw, dk):
    """uq_pack_stats: with the right guess the tables AND the statistics equal those of the two separate passes -- and the oracle's."""
    spec = synth.Spec(S + 6, length, **kw)
    host = synth.fastq_array(spec, n)
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, n)
    hs = ops.stats_fetch(ctx, st)
    ref = oracle_c.stats(host, oracle_c.index_lines(host), 0, n)
    assert np.array_equal(hs.counts, ref['counts'])
    d = _decide_from_stats(hs, **dk)
    if dk.get('notricks'): assert d['bits_per_base'] == 3 and d['bases'] == 'ACGNT'
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
    ref_d, ref_q, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
    res = ops.pack_stats(ctx, d_buf, ls, 0, n, p)
    assert res is not None, 'this geometry (2- or 3-bit bases, contiguous qualities) has a fused kernel'
    dna, qual, bad2, st2 = res
    hs2 = ops.stats_fetch(ctx, st2)
    assert not hs2.incomplete
    assert np.array_equal(hs2.counts, hs.counts)
    assert (hs2.len_min, hs2.len_max, hs2.max_record_bytes, hs2.bad_plus, hs2.bad_len) == (hs.len_min, hs.len_max, hs.max_record_bytes, None, None)
    assert ops.bad_index(bad2) is None
    assert np.array_equal(ctx.to_numpy(dna), ctx.to_numpy(ref_d)) and np.array_equal(ctx.to_numpy(qual), ctx.to_numpy(ref_q))
    assert ops.same_pack_params(p, p)


DEFAULT_FORM_CASES = FUSED_CASES + [
    ('var_long_tail', 6000, (36, 301), dict(n_rate=1), {}),                       # (several tiles per workgroup: the LDS image is cleared by the copy-out, tile after tile)
    ('var_mostly_short', 4000, (1, 40), dict(n_rate=1), {}),
    ('var_two_lengths', 3000, (149, 150), {}, {}),                                # rows with one empty slot at most
    ('fixed_len8', 3000, 8, {}, {}), ('var_9_to_17', 3000, (9, 17), dict(n_rate=2), dict(notricks=True)),
]


@pytest.mark.parametrize('qn', [False, True], ids=['plain', 'qname-phase'])
@pytest.mark.parametrize('name,n,length,kw,dk', DEFAULT_FORM_CASES, ids=[c[0] for c in DEFAULT_FORM_CASES])
def test_default_encode_form_against_the_oracle(ctx, name, n, length, kw, dk, qn):
    """The kernel the product runs by default -- census lists instead of a record index, pack + statistics (+ QNAME fields) in one launch,
    the variable-length instances that skip a row's empty groups -- DIRECTLY against the C oracle (test_pack_stats_fused
    compares HIP with HIP): packed rows == oracle_c.pack, counts and ranges == oracle_c.stats, QNAME columns == the oracle's."""
    spec = synth.Spec(S + 61, length, **kw)
    host = synth.fastq_array(spec, n)
    hls = oracle_c.index_lines(host)
    ref = oracle_c.stats(host, hls, 0, n)
    d = _decide_from_stats(type('H', (), dict(counts=ref['counts'], len_min=ref['len_min'], len_max=ref['len_max'], nz_keys=None))(), **dk)
    rd, rq, rbad = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                                 d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    assert rbad is None
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], ref['max_record_bytes'], avg_record_bytes=host.size // n)
    for rep in range(2):                                                          # twice: the second launch starts from the first one's LDS
        ops.scribble_lds(ctx, 0x5EED0000 + rep) if rep == 0 else None
        d_buf = ctx.to_device(host)
        cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
        fq = None
        if qn:
            fq = ops.FusedQname(ctx, n + 7)
            ops.qname_guess_async(ctx, d_buf, None, fq)
        q = ops.pack_stats_async(ctx, d_buf, None, n + 7, p, fq=fq)
        assert q is not None
        if qn: ops.qname_fused_finish(ctx, fq)
        hs = ops.stats_fetch(ctx, q[3])
        nl, ok = cen.wait()
        if not ok: pytest.skip('more than 1024 newlines in a census tile: the queued form stands down (covered elsewhere)')
        assert nl == 4 * n and not hs.incomplete
        assert np.array_equal(hs.counts, ref['counts']) and (hs.len_min, hs.len_max, hs.max_record_bytes) == (ref['len_min'], ref['len_max'], ref['max_record_bytes'])
        assert np.array_equal(ctx.to_numpy(q[0][:n * d['dna_bytes_per_row']]).reshape(n, -1), rd), 'DNA rows differ from the oracle'
        assert np.array_equal(ctx.to_numpy(q[1][:n * d['quality_bytes_per_row']]).reshape(n, -1), rq), 'QUAL rows differ from the oracle'
        assert ops.bad_index(q[2]) is None
        if qn:
            from uq_amd import qname_device
            got = qname_device.analyse_fused(ctx, fq, n)
            lines = O.read_lines(host.tobytes())
            p1 = O.pass1(lines)
            ocols = O.qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
            oarr = O.qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], ocols)
            assert got is not None and list(got[:3]) == [p1['prefix'], p1['suffix'], p1['separators']] and got[3] == ocols
            for a, b in zip(got[4], oarr): assert np.array_equal(ctx.to_numpy(a, b.dtype), b)


def test_pack_stats_wrong_guesses(ctx):
    """A wrong guess never goes unnoticed: the statistics stay exact (or say they are incomplete) and differ from the guess."""
    n = 4000
    host = synth.fastq_array(synth.Spec(S + 7, (36, 301), n_rate=1), n)
    d_buf = ctx.to_device(host)
    nlines, ls = _index(ctx, d_buf)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, ls, 0, n)
    hs = ops.stats_fetch(ctx, st)
    d = _decide_from_stats(hs)
    mk = lambda dd, rec: ops.make_pack_params(dd['bases'], dd['qualities'], dd['N_qual'], dd['bits_per_base'], dd['bits_per_quality'],
                                              dd['variable_read_lengths'], dd['dna_bytes_per_row'], dd['quality_bytes_per_row'], dd['dna_max'], rec)
    truth = mk(d, hs.max_record_bytes)
    # (a) the guess allows for shorter reads than the file holds: counts flagged incomplete
    short = dict(d); short['dna_max'] = 200
    lv = 201
    short['dna_bytes_per_row'] = -(-d['bits_per_base'] * lv // 8); short['quality_bytes_per_row'] = -(-d['bits_per_quality'] * lv // 8)
    res = ops.pack_stats(ctx, d_buf, ls, 0, n, mk(short, hs.max_record_bytes))
    assert res is not None and ops.stats_fetch(ctx, res[3]).incomplete
    # (b) records longer than the guess' tile hint: incomplete as well
    res = ops.pack_stats(ctx, d_buf, ls, 0, n, mk(d, 300))
    assert res is not None and ops.stats_fetch(ctx, res[3]).incomplete
    # (c) a guessed quality alphabet that misses characters: the kernel counts on the codes of the guess, so a symbol without a code
    #     makes the counts incomplete (round 1's kernel counted characters and stayed exact: either is a wrong guess noticed)
    fewer = dict(d); fewer['qualities'] = d['qualities'][:-3]
    g = mk(fewer, hs.max_record_bytes)
    res = ops.pack_stats(ctx, d_buf, ls, 0, n, g)
    assert res is not None
    hs2 = ops.stats_fetch(ctx, res[3])
    assert hs2.incomplete or np.array_equal(hs2.counts, hs.counts)
    assert not ops.same_pack_params(g, truth)
    # (d) 3-bit DNA (--notricks keeps N as a base) has a fused kernel too: exact counts, the separate pass's tables
    d3 = _decide_from_stats(hs, notricks=True)
    p3 = mk(d3, hs.max_record_bytes)
    res = ops.pack_stats(ctx, d_buf, ls, 0, n, p3)
    assert res is not None
    hs3 = ops.stats_fetch(ctx, res[3])
    assert not hs3.incomplete and np.array_equal(hs3.counts, hs.counts)
    ref3 = ops.pack(ctx, d_buf, ls, 0, n, p3)
    assert ctx.torch.equal(res[0], ref3[0]) and ctx.torch.equal(res[1], ref3[1])
    # (e) a geometry without one -- eight bits per base (--pad on a five-letter alphabet): nothing launched
    d8 = _decide_from_stats(hs, notricks=True, pad=True)
    assert d8['bits_per_base'] == 4 and ops.pack_stats(ctx, d_buf, ls, 0, n, mk(d8, hs.max_record_bytes)) is None


@pytest.mark.parametrize('variable', [False, True])
def test_encoder_reference_signature(ctx, tmp_path, variable):
    """uq.encoder_fixed / encoder_variable called as the reference calls them (uq.py:707-708): ten positional arguments,
    `((dna_array, dna_ptr), (qual_array, qual_ptr), lib)` back, `lib.free(ptr)` afterwards (uq.py:712-713)."""
    import uq_oracle as O
    from uq_amd import uq
    n = 700
    fq = synth.fastq(20261003 + 61, n, (20, 75) if variable else 64, n_rate=2)
    path = tmp_path / 'in.fastq'; path.write_bytes(fq)
    lines = O.read_lines(fq)
    p1 = O.pass1(lines)
    d = O.decide(p1['static_qualities'], p1['dna_min'], p1['dna_max'])
    fn = uq.encoder_variable if variable else uq.encoder_fixed

    class Status: current = 0
    st = Status()
    (dna_array, dna_ptr), (qual_array, qual_ptr), lib = fn(n, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'],
                                                            d['quality_bytes_per_row'], st, str(path), d['bits_per_base'], d['bits_per_quality'], ctx=ctx)
    rd, rq = O.encoder(lines, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'], d['quality_bytes_per_row'],
                       d['bits_per_base'], d['bits_per_quality'], variable)
    assert tuple(dna_array.shape) == rd.shape and tuple(qual_array.shape) == rq.shape
    assert np.array_equal(dna_array.cpu().numpy(), rd) and np.array_equal(qual_array.cpu().numpy(), rq)
    assert dna_ptr == dna_array.data_ptr() and qual_ptr == qual_array.data_ptr() and st.current == n
    assert lib.free(dna_ptr) == 0 and lib.free(qual_ptr) == 0 and not lib._owned


@pytest.mark.parametrize('offset', [0, 7])
def test_chunked_census_equals_whole_buffer(ctx, offset):
    """the newline census taken chunk by chunk while a file streams into HBM (uq_count_lines_begin /
    _chunk / _end) gives the line count and -- through uq_index_lines -- the record index of the whole-buffer census,
    whatever the chunk sizes and their order."""
    t = ctx.torch
    src = ops.synth_fastq(ctx, synth.Spec(S + 80, (30, 120), n_rate=1), 0, 40_000)
    big = t.empty(src.numel() + 64, dtype=t.uint8, device=ctx.device).fill_(10)
    big[offset:offset + src.numel()] = src
    buf = big[offset:offset + src.numel()]
    nl = ops.count_lines(ctx, buf)
    ls = ops.index_lines(ctx, buf, nl)
    hls = oracle_c.index_lines(ctx.to_numpy(buf))
    assert nl == len(hls) - 1 and np.array_equal(ctx.to_numpy(ls, np.uint64), hls)        # the whole-buffer census against the oracle
    mis = buf.data_ptr() & 15
    tile = 16 << 10
    rng = np.random.default_rng(5)
    for trial in range(3):
        # cut points: multiples of 16 KiB in the aligned address space
        ntiles = (buf.numel() + mis + tile - 1) // tile
        cuts = sorted(set(rng.integers(1, ntiles, size=[1, 5, 40][trial]).tolist()))
        edges = [0] + [c * tile - mis for c in cuts] + [buf.numel()]
        pieces = [(a, b - a) for a, b in zip(edges[:-1], edges[1:]) if b > a]
        order = rng.permutation(len(pieces)) if trial else range(len(pieces))
        cc = ops.ChunkedCensus(ctx, buf)
        for k in order: cc.chunk(*pieces[k])
        assert cc.end() == nl
        assert t.equal(ops.index_lines(ctx, buf, nl), ls)
    # a chunk that is not made of whole tiles is refused
    from uq_amd._lib import UqHipError
    cc = ops.ChunkedCensus(ctx, buf)
    with pytest.raises(UqHipError):
        cc.chunk(0, 1000)


def test_session_load_overlaps_census_with_ingest(ctx, tmp_path):
    """Session.load: the census is queued behind every chunk's PCIe copy; index and statistics equal the whole-buffer path's."""
    from uq_amd import uq
    fq = synth.fastq(S + 81, 20_000, (40, 100), n_rate=1)
    p = tmp_path / 'in.fastq'; p.write_bytes(fq)
    args = uq.validate_args(uq.build_parser().parse_args(['-i', str(p), '--quiet']))
    a = uq.Session(args, ctx=ctx)
    a.io.chunk, a.io.nbuf = 64 << 10, 4                       # many chunks
    a.load(str(p))
    b = uq.Session(args, ctx=ctx)
    b.load_device(ctx.bytes_to_device(fq))
    assert a.total == b.total == 20_000 and ctx.torch.equal(a.d_ls, b.d_ls)
    assert np.array_equal(ops.stats_fetch(ctx, a.d_stats).counts, ops.stats_fetch(ctx, b.d_stats).counts)


@pytest.mark.parametrize('length,kw', [(150, {}), ((36, 301), dict(n_rate=1)), (100, dict(n_rate=2))], ids=['fixed150', 'var-ntrick', 'fixed100-ntrick'])
def test_queued_census_index_pack_equals_the_plain_calls(ctx, length, kw):
    """uq_count_lines_end_async + uq_index_lines_async + uq_pack_stats_async (line count taken on the device, nothing waits for the
    host in between) against uq_count_lines + uq_index_lines + uq_pack_stats: same count, same index, same tables, same statistics;
    a capacity that is too small and a buffer that is not the census's are refused the documented way."""
    from uq_amd import analysis, synth
    t = ctx.torch
    n = 50_000
    d_buf = ops.synth_fastq(ctx, synth.Spec(20261004, length, **kw), 0, n)
    nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
    guess = ops.head_guess_indexed(ctx, d_buf, ls, n)
    ref = ops.pack_stats(ctx, d_buf, ls, 0, n, guess)
    assert ref is not None
    hs_ref = ops.stats_fetch(ctx, ref[3])
    for chunks in (1, 3):
        cen = ops.ChunkedCensus(ctx, d_buf)
        step = ((d_buf.numel() // chunks) // 16384 + 1) * 16384
        for lo in range(0, d_buf.numel(), step): cen.chunk(lo, min(step, d_buf.numel() - lo))
        cen.end_async()
        cap = n + 1000
        ls2 = ops.index_lines_async(ctx, d_buf, 4 * cap)
        got = ops.pack_stats_async(ctx, d_buf, ls2, cap, guess)
        assert got is not None
        hs = ops.stats_fetch(ctx, got[3])
        nl2, ok = cen.wait()
        assert ok and nl2 == nl
        assert t.equal(ls2[:nl + 1], ls)
        assert t.equal(got[0][:n * guess.dna_bytes_per_row], ref[0]) and t.equal(got[1][:n * guess.quality_bytes_per_row], ref[1])
        assert not hs.incomplete and np.array_equal(hs.counts, hs_ref.counts)
        assert (hs.len_min, hs.len_max, hs.max_record_bytes) == (hs_ref.len_min, hs_ref.len_max, hs_ref.max_record_bytes)
    # the QNAME form without an expanded index (line_start = None): the pack kernel and the QNAME sample walk the census's lists
    # (csrc/lines.h) -- same tables, same statistics, same field values as with the index
    fa, fb = ops.FusedQname(ctx, n), ops.FusedQname(ctx, n + 1000)
    ops.qname_guess(ctx, d_buf, ls, n, fa)
    ref_q = ops.pack_stats(ctx, d_buf, ls, 0, n, guess, fq=fa)
    assert ref_q is not None
    ops.qname_fused_finish(ctx, fa)
    qa = ops.qname_fused_fetch(ctx, fa)
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    ops.qname_guess_async(ctx, d_buf, None, fb)
    got = ops.pack_stats_async(ctx, d_buf, None, n + 1000, guess, fq=fb)
    assert got is not None
    ops.qname_fused_finish(ctx, fb)
    hs = ops.stats_fetch(ctx, got[3])
    assert cen.wait() == (nl, True)
    qb = ops.qname_fused_fetch(ctx, fb)
    assert t.equal(got[0][:n * guess.dna_bytes_per_row], ref[0]) and t.equal(got[1][:n * guess.quality_bytes_per_row], ref[1])
    assert not hs.incomplete and np.array_equal(hs.counts, hs_ref.counts)
    assert (hs.len_min, hs.len_max, hs.max_record_bytes) == (hs_ref.len_min, hs_ref.len_max, hs_ref.max_record_bytes)
    assert qa.ok and qb.ok and qa.flags == 0 and qb.flags == 0 and qa.nsep == qb.nsep and qa.nreads == qb.nreads == n
    assert bytes(qa.seps) == bytes(qb.seps) and (qa.plen, qa.slen) == (qb.plen, qb.slen)
    for c in range(qa.nsep + 1): assert t.equal(fa.column(c, n), fb.column(c, n))
    # ... and the form without the QNAME phase
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    got = ops.pack_stats_async(ctx, d_buf, None, n + 1000, guess)
    assert got is not None
    hs = ops.stats_fetch(ctx, got[3])
    assert cen.wait() == (nl, True)
    assert t.equal(got[0][:n * guess.dna_bytes_per_row], ref[0]) and t.equal(got[1][:n * guess.quality_bytes_per_row], ref[1])
    assert not hs.incomplete and np.array_equal(hs.counts, hs_ref.counts)
    with pytest.raises(Exception):                               # the plain form has no census to fall back on
        ops.pack_stats(ctx, d_buf, None, 0, n, guess)
    # a census of ANOTHER buffer in between overwrites the lists: the index-free forms refuse instead of walking them
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    assert cen.wait() == (nl, True)
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    other_buf = ops.synth_fastq(ctx, synth.Spec(77, 60), 0, 500)
    assert ops.count_lines(ctx, other_buf) == 2000
    with pytest.raises(Exception):
        ops.pack_stats_async(ctx, d_buf, None, n + 1000, guess)
    # tables / index too small for the file: flagged, nothing written beyond them
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    cap = n // 2
    ls3 = ops.index_lines_async(ctx, d_buf, 4 * cap)
    got = ops.pack_stats_async(ctx, d_buf, ls3, cap, guess)
    hs = ops.stats_fetch(ctx, got[3])
    nl3, ok = cen.wait()
    assert nl3 == nl and not ok and hs.incomplete
    assert t.equal(ls3[:4 * cap + 1], ls[:4 * cap + 1])
    # lines of three bytes: more than 1024 newlines per 16 KiB tile, the census's lists overflow.  The queued index and pack
    # kernel run before the host knows: they must stand down (ok False, statistics incomplete) without reading beyond a slot
    # or the buffer; the plain calls (bitmap form of the index) then give the oracle's index
    tiny = np.frombuffer(b'@a\nAC\n+\nII\n' * 30_000, dtype=np.uint8).copy()
    backing = ctx.empty(tiny.size)                           # exactly as large as the text: nothing readable behind it
    backing.copy_(t.from_numpy(tiny))
    cen = ops.ChunkedCensus(ctx, backing); cen.chunk(0, backing.numel()); cen.end_async()
    cap = 30_000 + 64
    ls4 = ops.index_lines_async(ctx, backing, 4 * cap)
    got = ops.pack_stats_async(ctx, backing, ls4, cap, guess)
    hs = ops.stats_fetch(ctx, got[3])
    nl4, ok = cen.wait()
    assert nl4 == 120_000 and not ok and hs.incomplete
    nl5 = ops.count_lines(ctx, backing)
    assert nl5 == nl4 and np.array_equal(ctx.to_numpy(ops.index_lines(ctx, backing, nl5), np.uint64), oracle_c.index_lines(tiny))
    # the queued calls belong to the census that was closed last
    other = ops.synth_fastq(ctx, synth.Spec(5, 50), 0, 100)
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    with pytest.raises(Exception):
        ops.index_lines_async(ctx, other, 1000)
    assert cen.wait()[0] == nl


@pytest.mark.parametrize('shape', ['ragged', 'long-reads', 'mixed', 'short-lines', 'offset-buffer'])
def test_pack_from_the_census_lists_on_awkward_streams(ctx, shape):
    """The queued QNAME form without an index (line_start = None) against the indexed form on streams that stress the walk through the
    census's lists: reads from a few bases to the longest a tile holds (census tiles with a newline or two, pack tiles packed piece by piece,
    places found by the per-piece search), a few hundred newlines per census tile (just under a list's capacity), and a buffer that
    does not start 16-byte aligned.  Same tables, same statistics, same QNAME field values -- or both forms flagged incomplete."""
    t = ctx.torch
    rng = np.random.default_rng(hash(shape) % 1000)
    if shape == 'ragged': lens = rng.integers(1, 400, 6000)
    elif shape == 'long-reads': lens = rng.integers(2500, 7800, 300)             # records of 5 - 16 KB: a pack tile holds one or two, often packed piece by piece
    elif shape == 'mixed': lens = np.where(rng.random(3000) < 0.03, rng.integers(3000, 7500, 3000), rng.integers(20, 300, 3000))
    elif shape == 'short-lines': lens = rng.integers(27, 35, 40000)          # ~ 78 bytes a record: ~ 840 newlines per 16 KiB tile, under a list's 1024
    else: lens = rng.integers(50, 260, 5000)
    recs = []
    for i, L in enumerate(lens):
        L = int(L)
        recs.append(b'@r%d:%d:%d\n' % (i % 4, i, int(rng.integers(0, 100000))) + bytes(rng.choice(np.frombuffer(b'ACGT', np.uint8), L)) + b'\n+\n' +
                    bytes(rng.integers(40, 75, L).astype(np.uint8)) + b'\n')
    text = np.frombuffer(b''.join(recs), dtype=np.uint8)
    n = len(lens)
    if shape == 'offset-buffer':
        back = ctx.empty(text.size + 16)
        d_buf = back[5:5 + text.size]
        d_buf.copy_(t.from_numpy(text.copy()))
    else:
        d_buf = ctx.to_device(text)
    nl = ops.count_lines(ctx, d_buf)
    assert nl == 4 * n
    ls = ops.index_lines(ctx, d_buf, nl)
    assert np.array_equal(ctx.to_numpy(ls, np.uint64), oracle_c.index_lines(text))
    guess = ops.head_guess_indexed(ctx, d_buf, ls, n)
    assert guess is not None
    guess.avg_record_bytes = int(text.size // n)
    fa, fb = ops.FusedQname(ctx, n), ops.FusedQname(ctx, n + 9)
    ops.qname_guess(ctx, d_buf, ls, n, fa)
    ref = ops.pack_stats(ctx, d_buf, ls, 0, n, guess, fq=fa)
    assert ref is not None
    ops.qname_fused_finish(ctx, fa); qa = ops.qname_fused_fetch(ctx, fa)
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    ops.qname_guess_async(ctx, d_buf, None, fb)
    got = ops.pack_stats_async(ctx, d_buf, None, n + 9, guess, fq=fb)
    assert got is not None
    ops.qname_fused_finish(ctx, fb)
    nl2, ok = cen.wait()
    qb = ops.qname_fused_fetch(ctx, fb)
    assert nl2 == nl
    ha, hb = ops.stats_fetch(ctx, ref[3]), ops.stats_fetch(ctx, got[3])
    assert ok and not ha.incomplete and not hb.incomplete    # (the guess saw every read: nothing may fall outside it; no list overflows)
    if True:
        assert t.equal(got[0][:n * guess.dna_bytes_per_row], ref[0]) and t.equal(got[1][:n * guess.quality_bytes_per_row], ref[1])
        assert np.array_equal(ha.counts, hb.counts) and (ha.len_min, ha.len_max, ha.max_record_bytes) == (hb.len_min, hb.len_max, hb.max_record_bytes)
        if qa.ok and qb.ok and qa.flags == 0 and qb.flags == 0:
            assert (qa.plen, qa.slen, qa.nsep, qa.nreads) == (qb.plen, qb.slen, qb.nsep, qb.nreads)
            for c in range(qa.nsep + 1): assert t.equal(fa.column(c, n), fb.column(c, n))


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('UQ_LIST_FUZZ_N', '40'))))        # UQ_LIST_FUZZ_N=2000 for a longer hunt
def test_pack_from_the_census_lists_fuzz(ctx, seed):
    """The CLI fuzz's random FASTQ files (alphabets of 2 - 14 bases, quality alphabets with gaps, fixed / variable lengths from 1 bp, QNAME
    families the reference copes with or refuses) through both queued QNAME forms: with the expanded index and from the census's lists.
    Where the pack kernel has a fused form for the file's alphabets: same tables, same statistics (complete or not), and -- when both
    QNAME guesses stand -- the same field values."""
    from test_gpu_e2e import _fuzz_case
    t = ctx.torch
    fq, _ = _fuzz_case(np.random.default_rng(770_000 + seed))
    text = np.frombuffer(fq, dtype=np.uint8)
    d_buf = ctx.to_device(text)
    nl = ops.count_lines(ctx, d_buf)
    n = nl // 4
    ls = ops.index_lines(ctx, d_buf, nl)
    guess = ops.head_guess_indexed(ctx, d_buf, ls, n)
    if guess is None: return                                  # (Q9 alphabets, malformed heads: no speculative kernel)
    fa, fb = ops.FusedQname(ctx, n), ops.FusedQname(ctx, n + 3)
    ops.qname_guess(ctx, d_buf, ls, n, fa)
    ref = ops.pack_stats(ctx, d_buf, ls, 0, n, guess, fq=fa)
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
    ops.qname_guess_async(ctx, d_buf, None, fb)
    got = ops.pack_stats_async(ctx, d_buf, None, n + 3, guess, fq=fb)
    assert (ref is None) == (got is None)
    if got is not None: ops.qname_fused_finish(ctx, fb)
    nl2, ok = cen.wait()
    assert nl2 == nl
    if ref is None: return                                    # no fused kernel for these alphabets
    if not ok:                                                # reads of a few bases: more than 1024 newlines in a census tile, the list form stands down
        assert ops.stats_fetch(ctx, got[3]).incomplete
        return
    qb = ops.qname_fused_fetch(ctx, fb)
    ops.qname_fused_finish(ctx, fa); qa = ops.qname_fused_fetch(ctx, fa)
    ha, hb = ops.stats_fetch(ctx, ref[3]), ops.stats_fetch(ctx, got[3])
    assert ha.incomplete == hb.incomplete
    cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()         # ... and the form without the QNAME phase
    plain = ops.pack_stats_async(ctx, d_buf, None, n + 3, guess)
    hp = ops.stats_fetch(ctx, plain[3])
    assert cen.wait() == (nl, True) and hp.incomplete == ha.incomplete
    if not ha.incomplete:
        assert t.equal(got[0][:n * guess.dna_bytes_per_row], ref[0]) and t.equal(got[1][:n * guess.quality_bytes_per_row], ref[1])
        assert t.equal(plain[0][:n * guess.dna_bytes_per_row], ref[0]) and t.equal(plain[1][:n * guess.quality_bytes_per_row], ref[1])
        assert np.array_equal(ha.counts, hb.counts) and (ha.len_min, ha.len_max, ha.max_record_bytes) == (hb.len_min, hb.len_max, hb.max_record_bytes)
        st = oracle_c.stats(text, oracle_c.index_lines(text), 0, n)
        assert np.array_equal(hb.counts, st['counts']) and np.array_equal(hp.counts, st['counts'])
    if qa.ok and qb.ok and qa.flags == 0 and qb.flags == 0:
        assert (qa.plen, qa.slen, qa.nsep, qa.nreads, bytes(qa.seps)) == (qb.plen, qb.slen, qb.nsep, qb.nreads, bytes(qb.seps))
        for c in range(qa.nsep + 1): assert t.equal(fa.column(c, n), fb.column(c, n))


@pytest.mark.parametrize('bases', [b'ACGNT', b'ACGTRYKM', b'ACGTacg', b'ACGTBDH', b'ACGTIJ'], ids=lambda b: b.decode())
def test_pack_three_bit_alphabets(ctx, bases):
    """3-bit base alphabets: the lookup-free pack path when three bits of the characters tell the bases apart ((c >> s) & 7 for
    some s: ACGNT, ACGTRYKM, ACGTacg ...), the table path when no shift does (ACGTIJ: I and J share every such index with a base ...);
    fixed and variable lengths, a stray character reported, against the oracle's pack."""
    rng = np.random.default_rng(len(bases))
    for variable in (False, True):
        n = 4000
        recs = []
        for i in range(n):
            L = int(rng.integers(1, 120)) if variable else 101
            seq = rng.choice(np.frombuffer(bases, np.uint8), L)
            q = rng.integers(40, 80, L).astype(np.uint8)
            recs.append(b'@r%d:%d\n' % (i, i % 7) + bytes(seq) + b'\n+\n' + bytes(q) + b'\n')
        host = np.frombuffer(b''.join(recs), dtype=np.uint8)
        d_buf = ctx.to_device(host)
        nlines, ls = _index(ctx, d_buf)
        st = ops.stats_new(ctx); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); hs = ops.stats_fetch(ctx, st)
        from uq_amd import analysis
        d = analysis.decide_from_stats(hs, notricks=True)
        assert d['bits_per_base'] == 3 and d['bases'] == ''.join(sorted(bases.decode()))
        p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                                 d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
        dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
        assert ops.bad_index(bad) is None
        hls = oracle_c.index_lines(host)
        rd, rq, _ = oracle_c.pack(host, hls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                                  d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
        assert np.array_equal(ctx.to_numpy(dna).reshape(n, -1), rd) and np.array_equal(ctx.to_numpy(qual).reshape(n, -1), rq)
    # a character outside the alphabet (one that shares its three bits with a base): named, not packed as that base
    bad_host = host.copy()
    at = int(hls[4 * 1234 + 1]) + 0
    bad_host[at] = ord('Z') if ord('Z') not in bases else ord('X')
    dna, qual, bad = ops.pack(ctx, ctx.to_device(bad_host), ls, 0, n, p)
    assert ops.bad_index(bad) == 1234


def test_side_context_allocations_follow_the_discipline(ctx):
    """SideContext (uq_amd/device.py): what head_guess allocates under scope() belongs to the side stream and what it adopts from the
    main stream is recorded on it -- two guesses and two whole queued steps back to back, WITHOUT a host wait between them, while
    the main stream allocates and frees blocks of the same sizes: the guesses and the tables stay those of the plain calls."""
    from uq_amd.device import SideContext
    t = ctx.torch
    side = SideContext(ctx)
    n = 60_000
    d_buf = ops.synth_fastq(ctx, synth.Spec(20261009, 100), 0, n)
    nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
    ref_guess = ops.head_guess_indexed(ctx, d_buf, ls, n)
    ref = ops.pack_stats(ctx, d_buf, ls, 0, n, ref_guess)
    ctx.sync()
    outs = []
    for it in range(2):
        st = ops.stats_new(ctx)
        cen = ops.ChunkedCensus(ctx, d_buf); cen.chunk(0, d_buf.numel()); cen.end_async()
        junk = [t.empty(512 * 1024, dtype=t.uint8, device=ctx.device).fill_(it) for _ in range(4)]     # main-stream blocks of the sizes the guess uses
        g, rpb = ops.head_guess(side, d_buf, head_bytes=ops.HEAD_BYTES_SMALL, head_reads=ops.HEAD_READS_INDEXED)
        del junk
        assert ops.same_pack_params(g, ref_guess)
        cap = n + 512
        ls2 = ops.index_lines_async(ctx, d_buf, 4 * cap)
        outs.append((cen, ops.pack_stats_async(ctx, d_buf, ls2, cap, g, st=st)))                         # no wait: the next iteration queues behind it
    for cen, got in outs:
        nl2, ok = cen.wait() if cen is outs[-1][0] else (nl, True)                                      # (only the last census is still open)
        assert ok and nl2 == nl
        assert t.equal(got[0][:n * g.dna_bytes_per_row], ref[0]) and t.equal(got[1][:n * g.quality_bytes_per_row], ref[1])
        assert np.array_equal(ops.stats_fetch(ctx, got[3]).counts, ops.stats_fetch(ctx, ref[3]).counts)
    # (no claim that the two streams differ: torch hands streams out of a pool of 32 per device, which wraps around in a long test session --
    # on one stream the two contexts merely serialise)
    assert side.stream.device == ctx.stream.device
