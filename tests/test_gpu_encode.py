"""GPU parity of the one-pass encoder (uq_encode_stream: census + record index + pass-1 statistics + speculative pack
in a single read of the stream) against the multi-pass entry points it replaces and against the oracle: whatever part of
its result the pass vouches for must be byte-identical; whatever it cannot vouch for it must say so (None)."""
import numpy as np
import pytest

import oracle_c
import uq_oracle as O
from uq_amd import analysis, ops, synth

pytestmark = pytest.mark.gpu
S = 20261003


def _multi_pass(ctx, d_buf):
    nl = ops.count_lines(ctx, d_buf)
    ls = ops.index_lines(ctx, d_buf, nl)
    st = ops.stats_new(ctx)
    if nl >= 4: ops.stats_accumulate(ctx, st, d_buf, ls, 0, nl // 4)
    return nl, ls, ops.stats_fetch(ctx, st)


def _params(ctx, d_buf, hs, nreads, notricks=False, **over):
    d = analysis.decide_from_counts(hs.counts, hs.len_min, hs.len_max, notricks=notricks)
    kw = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'], bits_per_quality=d['bits_per_quality'],
              variable=d['variable_read_lengths'], dna_bytes_per_row=d['dna_bytes_per_row'], quality_bytes_per_row=d['quality_bytes_per_row'],
              dna_max=d['dna_max'], max_record_bytes=hs.max_record_bytes, avg_record_bytes=d_buf.numel() // max(nreads, 1))
    kw.update(over)
    return d, ops.make_pack_params(**kw)


def _check_against_multipass(ctx, d_buf, slack=1.05, expect_tables=True):
    t = ctx.torch
    nl, ls, hs = _multi_pass(ctx, d_buf)
    n = nl // 4
    d, p = _params(ctx, d_buf, hs, n)
    e = ops.encode_stream(ctx, d_buf, p, int(n * slack) + 8)
    assert e is not None, 'no one-pass kernel for this geometry'
    assert e.nlines == nl
    assert e.line_start is not None and t.equal(e.line_start, ls), 'record index differs'
    assert e.stats is not None
    es = ops.stats_fetch(ctx, e.stats)
    assert np.array_equal(es.counts, hs.counts), 'pair counts differ'
    assert (es.len_min, es.len_max, es.max_record_bytes, es.bad_plus, es.bad_len) == (hs.len_min, hs.len_max, hs.max_record_bytes, hs.bad_plus, hs.bad_len)
    if expect_tables:
        assert e.tables is not None
        dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
        assert ops.bad_index(bad) is None
        assert t.equal(e.tables[0], dna), 'DNA rows differ from uq_pack'
        assert t.equal(e.tables[1], qual), 'QUAL rows differ from uq_pack'
    return e, d, p, ls, n


@pytest.mark.parametrize('n,length,kw', [(60_000, 150, {}), (40_000, (36, 301), dict(n_rate=1)), (30_000, 100, dict(n_rate=2)),
                                         (2_000, (1, 40), {}), (50_000, 36, {}), (7, 150, {}), (1, 400, {})],
                         ids=['fixed150', 'var36-301-ntrick', 'fixed100-ntrick', 'tiny-reads', 'short36', 'seven', 'one'])
def test_one_pass_equals_multi_pass(ctx, n, length, kw):
    """Right guess: index, statistics and both tables are the multi-pass path's, byte for byte (grid of 768 workgroups, so
    the larger cases run several tiles per workgroup and the look-back spans the whole grid)."""
    spec = synth.Spec(S + 70, length, **kw)
    d_buf = ops.synth_fastq(ctx, spec, 0, n)
    e, d, p, ls, n_ = _check_against_multipass(ctx, d_buf)
    assert n_ == n
    # and the oracle on a prefix of the reads (the C restatement of uq.py:108-254)
    k = min(n, 3000)
    host = ctx.to_numpy(d_buf[:int(ls[4 * k])])
    hls = oracle_c.index_lines(host)
    rd, rq, _ = oracle_c.pack(host, hls, 0, k, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                              d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    assert np.array_equal(ctx.to_numpy(e.tables[0][:k * d['dna_bytes_per_row']]).reshape(k, -1), rd)
    assert np.array_equal(ctx.to_numpy(e.tables[1][:k * d['quality_bytes_per_row']]).reshape(k, -1), rq)


@pytest.mark.parametrize('offset', [1, 5, 8, 15])
def test_one_pass_misaligned_buffer(ctx, offset):
    """The stream may start anywhere: tiles are cut on absolute 16-byte addresses."""
    t = ctx.torch
    src = ops.synth_fastq(ctx, synth.Spec(S + 71, (30, 90), n_rate=1), 0, 20_000)
    big = t.empty(src.numel() + 64, dtype=t.uint8, device=ctx.device).fill_(10)      # newlines all around: none may be counted
    big[offset:offset + src.numel()] = src
    _check_against_multipass(ctx, big[offset:offset + src.numel()])


def _fastq(records):
    return b''.join(b'@q:%d:%d\n%s\n+\n%s\n' % (i % 3, i, s, q) for i, (s, q) in enumerate(records))


def test_one_pass_flags_what_it_cannot_vouch_for(ctx):
    """Wrong guesses: the index and the statistics stay exact (they do not depend on the guess) unless the pass says
    otherwise, the tables are withdrawn; the caller's multi-pass fallback is then byte-identical to the plain path."""
    t = ctx.torch
    rng = np.random.default_rng(3)
    B, Q = np.frombuffer(b'ACGT', np.uint8), np.arange(40, 70, dtype=np.uint8)
    recs = [(bytes(rng.choice(B, 80)), bytes(rng.choice(Q, 80))) for _ in range(30_000)]
    good = ctx.bytes_to_device(_fastq(recs))
    nl, ls, hs = _multi_pass(ctx, good)
    d, guess = _params(ctx, good, hs, nl // 4)

    def run(buf, g=guess, slack=1.05):
        nl2, ls2, hs2 = _multi_pass(ctx, buf)
        e = ops.encode_stream(ctx, buf, g, int(nl2 // 4 * slack) + 8)
        assert e is not None and e.nlines == nl2
        if e.line_start is not None: assert t.equal(e.line_start, ls2)
        if e.stats is not None:
            es = ops.stats_fetch(ctx, e.stats)
            assert np.array_equal(es.counts, hs2.counts) and (es.len_min, es.len_max, es.bad_plus, es.bad_len) == (hs2.len_min, hs2.len_max, hs2.bad_plus, hs2.bad_len)
        return e, hs2

    e, _ = run(good)
    assert e.tables is not None and e.stats is not None and e.line_start is not None
    # (1) one quality outside the guessed range, far into the file
    r2 = list(recs); r2[25_000] = (r2[25_000][0], b'~' + r2[25_000][1][1:])
    e, _ = run(ctx.bytes_to_device(_fastq(r2)))
    assert e.tables is None and e.stats is not None and e.line_start is not None
    # (2) a base outside ACGT (no N-trick in the guess)
    r2 = list(recs); r2[7] = (b'N' + r2[7][0][1:], r2[7][1])
    e, _ = run(ctx.bytes_to_device(_fastq(r2)))
    assert e.tables is None and e.stats is not None
    # (3) a read longer than the guess's dna_max: its symbols are not all visited -> the statistics are withdrawn too
    r2 = list(recs); r2[12_345] = (r2[12_345][0] + b'ACGTACGT', r2[12_345][1] + b'IIIIIIII')
    e, _ = run(ctx.bytes_to_device(_fastq(r2)))
    assert e.tables is None and e.stats is None and e.line_start is not None
    # (4) a record longer than the halo a tile sees (3 kB): same
    r2 = list(recs); r2[20_000] = (b'A' * 3000, b'I' * 3000)
    e, _ = run(ctx.bytes_to_device(_fastq(r2)))
    assert e.tables is None and e.stats is None and e.line_start is not None
    # (5) SEQ / QUAL lengths differ, third line without '+': withdrawn / reported through the statistics as usual
    r2 = list(recs); r2[100] = (r2[100][0], r2[100][1][:-3])
    e, hs2 = run(ctx.bytes_to_device(_fastq(r2)))
    assert e.tables is None and hs2.bad_len == 100
    bad_plus = _fastq(recs).replace(b'\n+\n', b'\n-\n', 5000).replace(b'\n-\n', b'\n+\n', 4999)
    e, hs2 = run(ctx.bytes_to_device(bad_plus))
    assert hs2.bad_plus == 4999 and e.stats is not None and ops.stats_fetch(ctx, e.stats).bad_plus == 4999
    # (6) fewer rows than reads: index and tables withdrawn, the line count is still the census
    e, _ = run(good, slack=0.5)
    assert e.line_start is None and e.tables is None
    # (7) no final newline, and a line count that is no multiple of 4: counted as `wc -l` counts, tables withdrawn
    e, _ = run(ctx.bytes_to_device(_fastq(recs)[:-1]))
    assert e.nlines == 4 * len(recs) - 1 and e.tables is None and e.line_start is not None
    e, _ = run(ctx.bytes_to_device(_fastq(recs) + b'@trailing garbage'))
    assert e.nlines == 4 * len(recs) and e.tables is not None and e.stats is not None
    # (8) more lines in a tile than its list holds (2-byte lines): only the count survives
    e = ops.encode_stream(ctx, ctx.bytes_to_device(b'A\n' * 200_000), guess, 60_000)
    assert e.nlines == 200_000 and e.line_start is None and e.tables is None
    # (9) geometries without a one-pass kernel: 3-bit DNA, the Q9 carry code, records beyond the halo in the guess
    _, p3 = _params(ctx, good, hs, nl // 4, bases='ACGNT', bits_per_base=3, dna_bytes_per_row=30)
    assert ops.encode_stream(ctx, good, p3, 40_000) is None
    _, pl = _params(ctx, good, hs, nl // 4, max_record_bytes=5000)
    assert ops.encode_stream(ctx, good, pl, 40_000) is None


def test_one_pass_full_size_configs1(ctx):
    """BASELINE configs[1] (10 M x 150 bp): the one-pass result equals the multi-pass one (torch.equal on every output)."""
    d_buf = ops.synth_fastq(ctx, synth.Spec(S + 2, 150), 0, 10_000_000)
    _check_against_multipass(ctx, d_buf, slack=1.01)
