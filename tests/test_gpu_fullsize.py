"""GPU, BASELINE.json's full single-GPU size (configs[1]: 10 M x 150 bp = 3.4 GB): size-independent
properties instead of an oracle run -- encode -> decode round trip on the device, histogram mass,
index monotonicity, sortedness and key/permutation consistency of the table builds."""
import numpy as np
import pytest

from uq_amd import analysis, ops, synth

pytestmark = pytest.mark.gpu
N = 10_000_000
L = 150


@pytest.fixture(scope='module')
def packed(ctx):
    spec = synth.Spec(20261003 + 2, L)
    d_buf = ops.synth_fastq(ctx, spec, 0, N)
    nlines = ops.count_lines(ctx, d_buf)
    assert nlines == 4 * N
    ls, st = ops.index_and_stats(ctx, d_buf, nlines)
    hs = ops.stats_fetch(ctx, st)
    d = analysis.decide_from_counts(hs.counts, hs.len_min, hs.len_max)
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
    dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, N, p)
    assert ops.bad_index(bad) is None
    return dict(buf=d_buf, ls=ls, hs=hs, d=d, dna=dna, qual=qual)


def test_index_and_histogram_properties(ctx, packed):
    t = ctx.torch
    ls, hs, d = packed['ls'], packed['hs'], packed['d']
    assert int(ls[0]) == 0 and int(ls[-1]) == packed['buf'].numel()
    assert bool((ls[1:] > ls[:-1]).all())                                   # strictly increasing line starts
    assert bool((packed['buf'][(ls[1:] - 1)] == 10).all())                  # every line ends in '\n'
    assert int(hs.counts.sum()) == N * L                                     # one count per (base, quality) pair
    assert (hs.len_min, hs.len_max) == (L, L) and hs.bad_plus is None and hs.bad_len is None
    assert d['bases'] == 'ACGT' and d['bits_per_base'] == 2 and d['bits_per_quality'] == 6 and len(d['qualities']) == 41
    assert hs.max_record_bytes == int((ls[4::4] - ls[:-4:4]).max())


def test_pack_unpack_roundtrip_full_size(ctx, packed):
    """encode -> decode gives back every base and quality character of all 10 M reads."""
    t = ctx.torch
    d = packed['d']
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'], bits_per_quality=d['bits_per_quality'],
               variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'])
    seq, qt, ln, bad = ops.unpack(ctx, packed['dna'], packed['qual'], N, ops.make_unpack_params(cfg))
    assert ops.bad_index(bad) is None
    assert bool((ln == L).all())
    S = seq.view(N, L); Q = qt.view(N, L)
    ls, buf = packed['ls'], packed['buf']
    ar = t.arange(L, device=ctx.device)
    CH = 500_000
    for a in range(0, N, CH):
        b = min(N, a + CH)
        s0 = ls[4 * a + 1:4 * b + 1:4]; q0 = ls[4 * a + 3:4 * b + 3:4]
        assert t.equal(buf[s0[:, None] + ar], S[a:b]), 'sequence mismatch in reads %d..%d' % (a, b)
        assert t.equal(buf[q0[:, None] + ar], Q[a:b]), 'quality mismatch in reads %d..%d' % (a, b)


def test_table_build_properties_full_size(ctx, packed):
    """argsort / gather / unique on the 10 M x 38 B DNA table: sortedness, permutation, key consistency, and
    pattern -> inverse pattern round trips for all eight layouts."""
    t = ctx.torch
    C = packed['d']['dna_bytes_per_row']
    dna = packed['dna']
    perm, key, skey, uniq, nu = ops.unique_rows(ctx, dna, N, C)
    p64 = perm.to(t.int64) & 0xFFFFFFFF
    assert int(t.bincount(p64, minlength=N).max()) == 1                      # a permutation
    g = ops.gather_rows(ctx, dna, N, C, perm).view(N, C)
    a, b = g[:-1], g[1:]
    neq = a != b
    first = neq.to(t.uint8).argmax(dim=1)
    rows = t.arange(N - 1, device=ctx.device)
    ok = (~neq.any(dim=1)) | (a[rows, first] < b[rows, first])
    assert bool(ok.all())                                                     # memcmp-sorted
    k = (key.to(t.int64) & 0xFFFFFFFF)
    assert t.equal(uniq.view(nu, C)[k], dna.view(N, C))                       # table[key] reproduces every row (uq.py:953)
    sk = skey.to(t.int64) & 0xFFFFFFFF
    assert t.equal(sk, k[p64]) and int(sk[0]) == 0 and int(sk[-1]) == nu - 1 and bool(((sk[1:] - sk[:-1]) >= 0).all())
    # ties keep file order (stable): within equal keys the permutation is increasing
    same = sk[1:] == sk[:-1]
    assert bool((p64[1:][same] > p64[:-1][same]).all())
    for pat in ('0.1', '0.2', '1.1', '1.2', '2.1', '2.2', '3.1', '3.2'):
        pay = ops.pattern(ctx, dna, N, C, pat)
        assert t.equal(ops.unpattern(ctx, pay, N, C, pat), dna), pat
    # two layouts checked against their closed forms on a strided sample
    pay = ops.pattern(ctx, dna, N, C, '0.2')
    r = t.arange(0, N, 9973, device=ctx.device)
    assert t.equal(pay.view(C, N)[:, r].T.contiguous(), dna.view(N, C)[r])
    pay = ops.pattern(ctx, dna, N, C, '3.2')
    assert t.equal(pay.view(N, C)[N - 1 - r], dna.view(N, C)[r])


@pytest.mark.parametrize('with_qname', [True, False], ids=['qname', 'plain'])
def test_default_encode_form_beyond_2_gib(ctx, packed, with_qname):
    """The product's default step -- queued census, no record index, pack + statistics (+ QNAME fields) in one kernel -- over the 3.4 GB buffer: offsets
    beyond 2^31 reach the kernel through its tile records (round 4: a sign-extended `v_readlane` of such an offset faulted here and on nothing smaller).
    Tables and statistics must be those of the indexed plain kernels."""
    from uq_amd.device import SideContext
    t = ctx.torch
    d_buf, hs = packed['buf'], packed['hs']
    side = SideContext(ctx)
    st = ops.stats_new(ctx)
    census = ops.ChunkedCensus(ctx, d_buf); census.chunk(0, d_buf.numel()); census.end_async()
    guess, rpb = ops.head_guess(side, d_buf, head_bytes=ops.HEAD_BYTES_SMALL, head_reads=ops.HEAD_READS_INDEXED)
    cap = int(d_buf.numel() * rpb * 1.02) + 1024
    guess.avg_record_bytes = int(1.0 / rpb)
    fq = None
    if with_qname:
        fq = ops.FusedQname(ctx, cap); ops.qname_guess_async(ctx, d_buf, None, fq)
    sp = ops.pack_stats_async(ctx, d_buf, None, cap, guess, st=st, fq=fq)
    nlines, ok = census.wait()
    assert ok and nlines == 4 * N and sp is not None
    assert t.equal(sp[0][:N * guess.dna_bytes_per_row], packed['dna']) and t.equal(sp[1][:N * guess.quality_bytes_per_row], packed['qual'])
    h2 = ops.stats_fetch(ctx, sp[3])
    assert not h2.incomplete and np.array_equal(h2.counts, hs.counts) and (h2.len_min, h2.len_max, h2.max_record_bytes) == (hs.len_min, hs.len_max, hs.max_record_bytes)
    if with_qname:
        from uq_amd import qname_device
        ops.qname_fused_finish(ctx, fq)
        got = qname_device.analyse_fused(ctx, fq, N)
        want = qname_device.analyse_device(ctx, d_buf, packed['ls'], N)
        assert got is not None and want is not None and got[:4] == want[:4]
        for a, b in zip(got[4], want[4]): assert t.equal(a, b)
