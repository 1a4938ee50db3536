"""GPU, the north_star's own sizes (BASELINE.json configs[2], configs[4] and the 200 M x 150 bp single-GPU target):
the whole encode (`Session.load_device` -> analyse -> pack -> run_mix, i.e. the CLI minus file I/O) on FASTQ generated
in HBM, checked through size-independent properties -- the oracle cannot run at these sizes:
  * decode(encode(x)) == x, every byte of the FASTQ text, on the device;
  * the stored order is THE stable memcmp argsort (a permutation, rows non-decreasing, ties in file order: those three
    properties have exactly one solution) and every member equals the original table in that order;
  * unique tables strictly increasing, keys dense, table[key] == rows;
  * pattern -> inverse pattern is the identity for all eight layouts, two layouts against their closed forms.
UQ_NORTHSTAR_SCALE=k divides the read counts by k (rehearsal on a busy box); the suite runs them at 1."""
import os

import numpy as np
import pytest

from uq_amd import ops, synth, uq

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get('UQ_NORTHSTAR_SCALE', '1'))
PATTERNS = ('0.1', '0.2', '1.1', '1.2', '2.1', '2.2', '3.1', '3.2')


def _session(ctx, flags):
    args = uq.build_parser().parse_args(['-i', os.path.abspath(__file__), '--quiet'] + flags)   # -i only has to be a file
    uq.validate_args(args)
    return uq.Session(args, ctx=ctx)


def _free(ctx):
    import gc
    gc.collect()
    ctx.torch.cuda.empty_cache()


def _u(t_):
    """device integer tensor (any width, unsigned bit pattern) -> int64 values"""
    w = t_.element_size()
    v = t_.to(__import__('torch').int64)
    return v if w == 8 else v & ((1 << (8 * w)) - 1)


def _payload(s, name):
    return s.members[name][1]


def _rows_nondecreasing(t, g, strict=False, chunk=8_000_000):
    """memcmp order of consecutive rows of the (n, C) uint8 tensor g; returns (ok, equal-to-predecessor mask [n-1])"""
    n = g.shape[0]
    same = t.empty(n - 1, dtype=t.bool, device=g.device)
    ok = True
    for a in range(0, n - 1, chunk):
        b = min(n - 1, a + chunk)
        x, y = g[a:b], g[a + 1:b + 1]
        neq = x != y
        first = neq.to(t.uint8).argmax(dim=1)
        r = t.arange(b - a, device=g.device)
        eq = ~neq.any(dim=1)
        same[a:b] = eq
        less = x[r, first] < y[r, first]
        ok = ok and bool(((less & ~eq) | (eq & (not strict))).all())
    return ok, same


def _check_stable_argsort(t, table, n, C, perm):
    p64 = _u(perm)
    assert int(t.bincount(p64, minlength=n).max()) == 1 and int(p64.max()) == n - 1          # a permutation
    g = table.view(n, C)[p64]
    ok, same = _rows_nondecreasing(t, g)
    assert ok, 'rows not in memcmp order'
    assert bool((p64[1:][same] > p64[:-1][same]).all()), 'ties not in file order'
    return p64, g, same


def test_configs2_50M_sort_dna_keyed_and_raw(ctx):
    """BASELINE configs[2]: 50 M x 150 bp, --sort DNA + unique/index (keyed default) and --sort DNA --raw DNA QUAL QNAME,
    10 % of the reads copy one of N/16 templates."""
    t = ctx.torch
    N = 50_000_000 // SCALE
    T = max(1, N // 16)
    spec = synth.Spec(20261003 + 3, 150, dup='dna', dup_templates=T)
    buf = ops.synth_fastq(ctx, spec, 0, N)

    # ---- keyed (the reference's default): DNA + DNA.key, QUAL + QUAL.key, QNAME.key + QNAME_i
    s = _session(ctx, ['--sort', 'DNA'])
    s.load_device(buf)
    s.encode_loaded(write=False)
    assert s.total == N and s.d['bits_per_base'] == 2 and s.d['bits_per_quality'] == 6
    Cd, Cq = s.d['dna_bytes_per_row'], s.d['quality_bytes_per_row']
    assert (Cd, Cq) == (38, 113)
    dna, qual = s.tables['DNA'][0], s.tables['QUAL'][0]
    cols = s.tables['QNAME']
    perm = ops.argsort_rows(ctx, dna, N, Cd)
    p64, g, same = _check_stable_argsort(t, dna, N, Cd, perm)
    nu = int((~same).sum()) + 1
    # the duplicate rule: distinct rows = non-duplicates + distinct templates drawn (T (1 - (1 - 1/T)^K), K ~ N/10 draws)
    expect = 0.9 * N + T * (1 - np.exp(-0.1 * N / T))
    assert abs(nu - expect) < 0.002 * N + 50, (nu, expect)
    U = _payload(s, 'DNA').view(-1, Cd)
    assert U.shape[0] == nu
    ok, usame = _rows_nondecreasing(t, U, strict=True)
    assert ok and not bool(usame.any()), 'unique DNA rows not strictly increasing'
    kd = _u(_payload(s, 'DNA.key'))
    assert _payload(s, 'DNA.key').element_size() == ops.key_itemsize(nu - 1) == 4
    assert int(kd[0]) == 0 and int(kd[-1]) == nu - 1
    step = kd[1:] - kd[:-1]
    assert bool(((step == 0) | (step == 1)).all()) and bool(((step == 0) == same).all())     # dense ranks, stored sorted (uq.py:798)
    assert t.equal(U[kd], g)                                                                  # table[key] == sorted rows (uq.py:953)
    del g, U, kd, step
    # the other tables follow the DNA order (uq.py:743-745): QUAL rows and QNAME columns of read perm[j] at position j
    UQ = _payload(s, 'QUAL').view(-1, Cq)
    kq = _u(_payload(s, 'QUAL.key'))
    ok, usame = _rows_nondecreasing(t, UQ, strict=True)
    assert ok and not bool(usame.any())
    for a in range(0, N, 10_000_000):
        b = min(N, a + 10_000_000)
        assert t.equal(UQ[kq[a:b]], qual.view(N, Cq)[p64[a:b]]), 'QUAL[QUAL.key] != qual[sort_order]'
    del UQ, kq
    kn = _u(_payload(s, 'QNAME.key'))
    for i, c in enumerate(cols):
        uc = _u(_payload(s, 'QNAME_%d' % (i + 1)))
        assert t.equal(uc[kn], _u(c)[p64]), 'QNAME column %d' % (i + 1)
    del kn, uc
    s.members = {}
    del s
    _free(ctx)

    # ---- raw, with two non-trivial layouts
    s = _session(ctx, ['--sort', 'DNA', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '2.2', '3.1'])
    s.load_device(buf)
    s.encode_loaded(write=False)
    assert t.equal(s.tables['DNA'][0], dna) and t.equal(s.tables['QUAL'][0], qual)            # same pack, same bytes
    pd, pq = _payload(s, 'DNA.raw'), _payload(s, 'QUAL.raw')
    assert t.equal(ops.unpattern(ctx, pd, N, Cd, '2.2').view(N, Cd), dna.view(N, Cd)[p64])
    sq = ops.unpattern(ctx, pq, N, Cq, '3.1').view(N, Cq)
    for a in range(0, N, 10_000_000):
        b = min(N, a + 10_000_000)
        assert t.equal(sq[a:b], qual.view(N, Cq)[p64[a:b]])
    # closed forms on a strided sample: 2.2 = reverse of the transposed stream, 3.1: out[i*R + j] = T[R-1-j][i]
    r = t.arange(0, N, 9973, device=ctx.device)
    assert t.equal(pd.flip(0).view(Cd, N)[:, r].T.contiguous(), dna.view(N, Cd)[p64[r]])
    assert t.equal(pq.view(Cq, N)[:, N - 1 - r].T.contiguous(), qual.view(N, Cq)[p64[r]])
    for i, c in enumerate(cols):
        assert t.equal(_u(_payload(s, 'QNAME_%d.raw' % (i + 1))), _u(c)[p64])
    s.members = {}
    del s, sq, pd, pq
    _free(ctx)


@pytest.mark.parametrize('mode', ['ntrick', 'notricks'])
def test_configs4_100M_variable_length(ctx, mode):
    """BASELINE configs[4]: 100 M reads of 36-301 bp with 1 % N (N always '!', nothing else '!'): default = N-trick,
    2-bit DNA; --notricks = ACGNT, 3-bit DNA.  Round trip of the whole text + all eight pattern ids, both tables."""
    t = ctx.torch
    N = 100_000_000 // SCALE
    spec = synth.Spec(20261003 + 5, (36, 301), n_rate=1)
    buf = ops.synth_fastq(ctx, spec, 0, N)
    s = _session(ctx, ['--raw', 'DNA', 'QUAL', 'QNAME'] + (['--notricks'] if mode == 'notricks' else []))
    s.load_device(buf)
    s.encode_loaded(write=False)
    d = s.d
    assert s.total == N and d['variable_read_lengths'] and (d['dna_min'], d['dna_max']) == (36, 301)
    assert int(s.hs.counts.sum()) == int(s.d_ls[2::4].sum() - s.d_ls[1::4].sum()) - N            # one count per base
    if mode == 'ntrick':
        assert d['bases'] == 'ACGT' and d['N_qual'] == {'N': 0} and d['bits_per_base'] == 2 and d['dna_bytes_per_row'] == 76
    else:
        assert d['bases'] == 'ACGNT' and d['N_qual'] == {} and d['bits_per_base'] == 3 and d['dna_bytes_per_row'] == 114
    assert d['bits_per_quality'] == 6 and d['quality_bytes_per_row'] == 227
    (dna, _, Cd), (qual, _, Cq) = s.tables['DNA'], s.tables['QUAL']
    s.d_ls = s.d_stats = None
    _free(ctx)
    # decode == input, every byte (uq.py:1002-1058 in one kernel)
    text = s.decode_text(dict(s.config), (dna, N, Cd), (qual, N, Cq), s.tables['QNAME'])
    assert text.numel() == buf.numel() and t.equal(text, buf), 'decode(encode(x)) != x'
    del text
    _free(ctx)
    # the --test sweep's transforms: all eight ids, both tables
    for tab, C in ((dna, Cd), (qual, Cq)):
        for pat in PATTERNS:
            pay = ops.pattern(ctx, tab, N, C, pat)
            back = ops.unpattern(ctx, pay, N, C, pat)
            assert t.equal(back, tab), (pat, C)
            if pat == '1.2':          # each row byte-reversed
                r = t.arange(0, N, 99991, device=ctx.device)
                assert t.equal(pay.view(N, C)[r].flip(1), tab.view(N, C)[r])
            del pay, back
    s.members = {}
    del s, dna, qual, buf
    _free(ctx)


def test_200M_x150_single_gpu_encode(ctx):
    """The north_star's target size on ONE GPU: 200 M x 150 bp (67.9 GB FASTQ), --sort None --raw DNA QUAL QNAME
    (configs[1]'s flags): encode, then decode every byte back on the device."""
    t = ctx.torch
    N = 200_000_000 // SCALE
    L = 150
    buf = ops.synth_fastq(ctx, synth.Spec(20261003 + 2, L), 0, N)
    s = _session(ctx, ['--raw', 'DNA', 'QUAL', 'QNAME'])
    s.load_device(buf)
    s.encode_loaded(write=False)
    d, hs, ls = s.d, s.hs, s.d_ls
    assert s.total == N and int(hs.counts.sum()) == N * L and (hs.len_min, hs.len_max) == (L, L)
    assert d['bases'] == 'ACGT' and d['bits_per_base'] == 2 and d['bits_per_quality'] == 6
    assert int(ls[0]) == 0 and int(ls[-1]) == buf.numel()
    for a in range(0, 4 * N, 100_000_000):                      # strictly increasing line starts, every line ends in '\n'
        b = min(4 * N, a + 100_000_000)
        assert bool((ls[a + 1:b + 1] > ls[a:b]).all()) and bool((buf[ls[a + 1:b + 1] - 1] == 10).all())
    assert _payload(s, 'DNA.raw').numel() == N * 38 and _payload(s, 'QUAL.raw').numel() == N * 113
    (dna, _, Cd), (qual, _, Cq) = s.tables['DNA'], s.tables['QUAL']
    # the queued step without a record index (bench.py's step: the pack kernel and the QNAME sample walk the census's newline lists; 4.1 M
    # census tiles, positions beyond 2^32): the same tables, every QNAME field value what the session's columns hold
    guess = ops.head_guess_indexed(ctx, buf, ls, N)
    cen = ops.ChunkedCensus(ctx, buf); cen.chunk(0, buf.numel()); cen.end_async()
    fq = ops.FusedQname(ctx, N + 1000)
    ops.qname_guess_async(ctx, buf, None, fq)
    got = ops.pack_stats_async(ctx, buf, None, N + 1000, guess, fq=fq)
    assert got is not None
    ops.qname_fused_finish(ctx, fq)
    assert cen.wait() == (4 * N, True)
    hq = ops.stats_fetch(ctx, got[3])
    assert not hq.incomplete and np.array_equal(hq.counts, hs.counts)
    assert t.equal(got[0][:N * Cd], dna.view(-1)[:N * Cd]) and t.equal(got[1][:N * Cq], qual.view(-1)[:N * Cq])
    from uq_amd import qname_device
    qres = qname_device.analyse_fused(ctx, fq, N)
    assert qres is not None and list(qres[:3]) == [s.config['QNAME_prefix'], s.config['QNAME_suffix'], s.config['QNAME_separators']]
    assert qres[3] == s.config['QNAME_columns']
    del got, fq, qres, cen
    s.d_ls = s.d_stats = ls = None
    _free(ctx)
    text = s.decode_text(dict(s.config), (dna, N, Cd), (qual, N, Cq), s.tables['QNAME'])
    assert text.numel() == buf.numel() and t.equal(text, buf), 'decode(encode(x)) != x'
    s.members = {}
    del text, s, dna, qual, buf
    _free(ctx)


def test_configs3_shard_25M_sort_qual_through_the_exchange_code(ctx):
    """BASELINE configs[3] is 200 M x 150 bp over 8 GPUs, `--sort QUAL --raw DNA QUAL QNAME`: one rank's share is 25 M reads.  That
    shard, at its real size, through the code the sharded encoder runs (uq_amd.dist.global_sort_rows + dist_gather_rows on HipRows;
    one rank: no exchange partner on a one-GPU box), 10 % of the reads copying one of N/16 QUAL templates: the order that comes
    back is THE stable memcmp argsort of the 113-byte QUAL rows and the DNA rows follow it."""
    from uq_amd import analysis, dist as uqdist
    t = ctx.torch
    N = 25_000_000 // SCALE
    spec = synth.Spec(20261003 + 4, 150, dup='qual', dup_templates=max(1, N // 16))
    buf = ops.synth_fastq(ctx, spec, 0, N)
    nl = ops.count_lines(ctx, buf)
    ls, st = ops.index_and_stats(ctx, buf, nl)
    hs = ops.stats_fetch(ctx, st)
    d = analysis.decide_from_stats(hs)
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
    dna, qual, bad = ops.pack(ctx, buf, ls, 0, N, p)
    assert ops.bad_index(bad) is None
    del buf, ls
    Cd, Cq = d['dna_bytes_per_row'], d['quality_bytes_per_row']
    assert (Cd, Cq) == (38, 113)
    be = uqdist.HipRows(ctx)
    gs = uqdist.global_sort_rows(be, qual, N, Cq, 0, total_rows=N)
    assert gs['rows'] == N and gs['offset'] == 0
    p64, g, same = _check_stable_argsort(t, qual, N, Cq, gs['gidx'].to(t.int32))
    assert t.equal(gs['table'].view(N, Cq), g)
    nu = int((~same).sum()) + 1
    T = max(1, N // 16)
    expect = 0.9 * N + T * (1 - np.exp(-0.1 * N / T))
    assert abs(nu - expect) < 0.002 * N + 50, (nu, expect)
    del g
    moved = uqdist.dist_gather_rows(be, dna, N, Cd, [0, N], gs['gidx'])
    for a in range(0, N, 10_000_000):
        b = min(N, a + 10_000_000)
        assert t.equal(moved.view(N, Cd)[a:b], dna.view(N, Cd)[p64[a:b]])
    # ... and back: rows scattered to file order by the same indices give the table again
    back = uqdist.dist_scatter_rows(be, moved, Cd, [0, N], gs['gidx'])
    assert t.equal(back, dna)
    _free(ctx)
