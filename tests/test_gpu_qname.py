"""GPU parity of the device QNAME passes (uq_qname_layout / uq_qname_tokenise / uq_prefix_distinct /
uq_encode_int + uq_amd.qname_device) against the oracle's sequential restatement of uq.py:394-444, 555-678,
717-736.  The device path may decline an input (returns None -> the host path runs); when it answers,
the answer must be the oracle's, and when it refuses, the oracle must refuse too."""
import numpy as np
import pytest

import uq_oracle as O
from uq_amd import ops, qname, qname_device, synth

pytestmark = pytest.mark.gpu


def _fastq(names):
    # (qualities that vary per base: with one quality for all of them every base would be an N-trick base, and the pack kernel the
    # fused QNAME pass rides in has no speculative form for the new quality codes of Q9)
    quals = (b'FGHI', b'GHIF', b'HIFG', b'IFGH')
    return b''.join(n + b'\nACGT\n+\n' + quals[i & 3] + b'\n' for i, n in enumerate(names))


def _oracle(fq):
    """('ok', prefix, suffix, separators, columns, arrays) or ('error', exception)."""
    lines = O.read_lines(fq)
    try:
        p1 = O.pass1(lines)
        cols = O.qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
        arr = O.qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], cols)
    except Exception as e:          # UqError, or the reference's own IndexError / re.error
        return ('error', e)
    return ('ok', p1['prefix'], p1['suffix'], p1['separators'], cols, arr)


def _index_gpu(ctx, fq):
    buf = ctx.to_device(np.frombuffer(fq, dtype=np.uint8))
    nlines = ops.count_lines(ctx, buf)
    return buf, ops.index_lines(ctx, buf, nlines), nlines // 4


INDEX = _index_gpu      # tests/test_qname_device_cpu.py swaps this (and qname_device.ops) for numpy stand-ins


def _device(ctx, fq):
    buf, ls, n = INDEX(ctx, fq)
    try:
        got = qname_device.analyse_device(ctx, buf, ls, n)
    except qname.QnameError as e:
        return ('error', e)
    if got is None:
        return ('declined',)
    pre, suf, sep, cols, arrs = got
    return ('ok', pre, suf, sep, cols, [ctx.to_numpy(a, np.dtype(c['dtype'])) for a, c in zip(arrs, cols)])


HUNT = int(__import__('os').environ.get('UQ_QNAME_HUNT', '0'))      # > 0: other seeds and ten times the cases in the random-grammar tests
FUSED = True            # tests/test_qname_device_cpu.py (numpy stand-ins, no pack kernel) switches the fused checks off
FUSED_TALLY = {'ok': 0, 'declined': 0}


def _fused(ctx, fq):
    """The QNAME passes inside the pack kernel (uq_qname_guess -> uq_pack_stats_qname -> uq_qname_fused_finish ->
    qname_device.analyse_fused): ('ok', ...) or ('declined',) -- the fused pass never refuses on its own, it stands down."""
    buf, ls, n = INDEX(ctx, fq)
    guess = ops.head_guess_indexed(ctx, buf, ls, n) if n else None
    if guess is None:
        return ('declined',)
    fq_dev = ops.FusedQname(ctx, n)
    ops.qname_guess(ctx, buf, ls, n, fq_dev)
    if ops.pack_stats(ctx, buf, ls, 0, n, guess, fq=fq_dev) is None:
        return ('declined',)
    ops.qname_fused_finish(ctx, fq_dev)
    got = qname_device.analyse_fused(ctx, fq_dev, n)
    if got is None:
        return ('declined',)
    pre, suf, sep, cols, arrs = got
    return ('ok', pre, suf, sep, cols, [ctx.to_numpy(a, np.dtype(c['dtype'])) for a, c in zip(arrs, cols)])


def _fused_from_census(ctx, fq):
    """The queued form without an expanded index (line_start = None): census -> uq_qname_guess_async -> uq_pack_stats_qname_async take the
    line starts from the census's newline lists (csrc/lines.h); the sample is stratified by position instead of by read number, so it may
    decline where _fused answers and the other way round -- an answer is the oracle's either way.  The tables must be the indexed form's."""
    buf, ls, n = INDEX(ctx, fq)
    guess = ops.head_guess_indexed(ctx, buf, ls, n) if n else None
    if guess is None:
        return ('declined',)
    cen = ops.ChunkedCensus(ctx, buf); cen.chunk(0, buf.numel()); cen.end_async()
    fq_dev = ops.FusedQname(ctx, n + 5)
    ops.qname_guess_async(ctx, buf, None, fq_dev)
    res = ops.pack_stats_async(ctx, buf, None, n + 5, guess, fq=fq_dev)
    if res is not None: ops.qname_fused_finish(ctx, fq_dev)
    nl, ok = cen.wait()
    assert nl == 4 * n
    if not ok:
        return ('declined', 'the lists overflowed')      # lines of a few bytes: more than 1024 newlines in a 16 KiB tile; the queued form stands down
    if res is None:
        return ('declined',)
    ref = ops.pack_stats(ctx, buf, ls, 0, n, guess)
    assert ctx.torch.equal(res[0][:n * guess.dna_bytes_per_row], ref[0]) and ctx.torch.equal(res[1][:n * guess.quality_bytes_per_row], ref[1])
    a, b = ops.stats_fetch(ctx, res[3]), ops.stats_fetch(ctx, ref[3])
    assert a.incomplete == b.incomplete and (a.incomplete or (np.array_equal(a.counts, b.counts) and (a.len_min, a.len_max, a.max_record_bytes) == (b.len_min, b.len_max, b.max_record_bytes)))
    got = qname_device.analyse_fused(ctx, fq_dev, n)
    if got is None:
        return ('declined',)
    pre, suf, sep, cols, arrs = got
    return ('ok', pre, suf, sep, cols, [ctx.to_numpy(a, np.dtype(c['dtype'])) for a, c in zip(arrs, cols)])


def _check_fused(ctx, fq, want, must_answer=False):
    """An answer of the fused pass must be the oracle's answer (so the oracle must HAVE one); anything else it declines.  Both forms:
    with the expanded index and straight from the census's lists."""
    got = _fused_from_census(ctx, fq)
    _check_fused_one(ctx, got, want, must_answer and len(got) == 1)
    _check_fused_one(ctx, _fused(ctx, fq), want, must_answer)


def _check_fused_one(ctx, got, want, must_answer=False):
    FUSED_TALLY[got[0]] += 1
    if got[0] == 'declined':
        assert not must_answer, 'the fused QNAME pass declined an input it is meant to handle'
        return
    assert want[0] == 'ok', ('the fused pass answered where the reference refuses', got[1:5], want)
    assert got[1:4] == want[1:4], (got[1:4], want[1:4])
    assert got[4] == want[4], (got[4], want[4])
    for a, b in zip(got[5], want[5]):
        assert a.dtype == b.dtype and np.array_equal(a, b)


def _check(ctx, fq, must_answer=False, fused_must_answer=False):
    want, got = _oracle(fq), _device(ctx, fq)
    if FUSED: _check_fused(ctx, fq, want, fused_must_answer)
    if got[0] == 'declined':
        assert not must_answer, 'device path declined an input it is meant to handle'
        return 'declined'
    assert got[0] == want[0], (got, want)
    if got[0] == 'ok':
        assert got[1:4] == want[1:4]
        assert got[4] == want[4]
        for a, b in zip(got[5], want[5]):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    return got[0]


def test_synthetic_illumina_names(ctx):
    for fq in (synth.fastq(5, 3000, 50), synth.fastq(6, 25000, 8)):
        assert _check(ctx, fq, must_answer=True, fused_must_answer=True) == 'ok'
    for n in (1, 2, 3, 11):
        _check(ctx, synth.fastq(7, n, 20), must_answer=True)      # tiny files: same answer or same refusal


def test_golden_fastq(ctx):
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    for f in sorted(os.listdir(gold)):
        if not f.endswith('.fastq') or f.endswith('.refdecode.fastq'): continue
        fq = open(os.path.join(gold, f), 'rb').read()
        if f.endswith('_refused.fastq'):
            # the reference refuses (its own words are in the json): the exact device path refuses or leaves it to the host path, which
            # refuses; the fused pass stands down (an answer would fail inside _check)
            assert _check(ctx, fq) in ('error', 'declined')
            continue
        # every written fixture is inside the device subset (no mapping strings beyond 8 bytes), except names with a regex
        # metacharacter among the separators ('-' in qn_seps_mixed): those the device path leaves to Python's `re` by design
        want = _oracle(fq)
        inside = not (set(want[3]) & qname_device.REGEX_SPECIAL)
        assert _check(ctx, fq, must_answer=inside) == ('ok' if inside else 'declined'), f


def test_mapping_columns_and_suffix(ctx):
    names = [b'@m%d#x:%d:%s#1' % (i % 7, i % 300, b'ab' if i % 3 else b'c') for i in range(500)]
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
    # a wide-range integer column with few distinct values stays a mapping of strings, sorted as text
    names = [b'@r:%d:%s' % ([5, 70000, 12345678, 31][i % 4], [b'left', b'right'][i % 2]) for i in range(400)]
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
    # negative numbers / explicit plus sign / offsets
    names = [b'@s_%d_%d' % (i % 50 - 25, 100000 + i) for i in range(300)]
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
    names = [b'@s;%s;%d' % ([b'+5', b'-5', b'5', b'05'][i % 4], i) for i in range(64)]
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'


def test_long_integer_fields(ctx):
    # SRA-style read numbers beyond 8 digits: 8-byte key falls back to the value (canonical integers)
    names = [b'@SRR1.%d/%d' % ((i * 7919317) % 900000000 + 100000000, i % 2 + 1) for i in range(12000)]
    assert _check(ctx, _fastq(names), must_answer=True, fused_must_answer=True) == 'ok'
    # ... but a column of long numbers that stays a mapping needs the strings themselves: host path
    names = [b'@r:%d:%d' % ([100000000000, 5][i % 2], i) for i in range(200)]
    assert _device(ctx, _fastq(names))[0] == 'declined'


def test_demotion_checkpoints(ctx):
    """mapping -> integers at 10 000 / 20 000 reads and at the end (uq.py:586-602, 634-638)."""
    # column 1: 1001 distinct values among the first 10 001 reads (just over 10000 // 10) -> integers at the checkpoint
    # column 2: exactly 1000 distinct there (stays mapping), 2001 among the first 20 001 -> integers at the second checkpoint
    # column 3: few distinct strings, never demoted
    n = 23000
    names = []
    for i in range(n):
        a = i % 1001
        b = i % 1000 if i <= 10000 else i % 2001
        names.append(b'@q:%d:%d:%s' % (a * 3, b * 1000, [b'x', b'yy', b'zzz'][i % 3]))
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
    # a column that crosses the threshold only at the final check (entries_read = n - 1)
    n = 15000
    names = [b'@q:%d:%d' % (i % 7, (i % 1000) if i < 12000 else i) for i in range(n)]
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
    # demotion of a column that holds a non-integer: "strings" -> refused, by both
    names = [b'@q:%d:%s' % (i % 7, b'abc' if i == 5 else b'%d' % i) for i in range(12000)]
    assert _check(ctx, _fastq(names)) == 'error'
    # the non-integer arrives after the demotion
    names = [b'@q:%d:%s' % (i % 7, b'abc' if i == 11000 else b'%d' % i) for i in range(12000)]
    assert _check(ctx, _fastq(names)) == 'error'


def test_refusals_and_declines(ctx):
    assert _check(ctx, _fastq([b'@r1', b'@r2', b'@r3'])) == 'error'                      # no separator survives (Q13)
    assert _device(ctx, _fastq([b'@a.1.x', b'@a.2.y', b'@a.3.z']))[0] == 'declined'      # regex metacharacter separator
    assert _device(ctx, _fastq([b'@r:%s%d:%d' % (b' ' if i % 2 else b'', i, i % 2) for i in range(50)]))[0] == 'declined'   # int() strips whitespace
    assert _device(ctx, _fastq([b'@r:%s:%d' % (b'averyveryverylongname' if i % 2 else b'short', i) for i in range(50)]))[0] == 'declined'
    assert _device(ctx, _fastq([b'@ab:1', b'@ab', b'@ab:2']))[0] in ('declined', 'error')  # a QNAME that is a proper prefix of line 1
    # A separator candidate that is line 1's last character before the suffix: in the reference's set (counted over the whole middle) but not
    # in the order (read from a slice one character short, Q14) -- the reference then errs or not by the LAST read alone.  Found by the CLI fuzz
    # (seed 700): the fused pass answered '=' where the reference gives up; it must leave these to the exact path.
    for names in ([b'@q_0=1', b'@q_1=4'], [b'@q_0=1', b'@q_1=4', b'@q_2=1'], [b'@q_%d=1' % (i % 7) for i in range(40)] + [b'@q_1=4'],
                  [b'@r:7:%d:7' % i for i in range(30)], [b'@r:%d:5' % (10 * i + 5) for i in range(30)]):
        _check(ctx, _fastq(names))
        if FUSED:
            got = _fused(ctx, _fastq(names))
            want = _oracle(_fastq(names))
            assert got[0] == 'declined' or want[0] == 'ok'


def _random_family(rng, n):
    """A random QNAME grammar: prefix, fields of random kinds joined by random separators, suffix."""
    seps_pool = ':_/#=;,@ '
    nf = int(rng.integers(2, 6))
    seps = [seps_pool[int(rng.integers(0, len(seps_pool) - 1))] for _ in range(nf - 1)]
    kinds = [['const', 'small', 'big', 'neg', 'word', 'pad', 'mixed'][int(rng.integers(0, 7))] for _ in range(nf)]
    vocab = [bytes(rng.choice(list(b'abcXYZ019'), size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(12)]
    prefix = [b'@', b'@RUN7', b'@x.y-'][int(rng.integers(0, 3))]
    suffix = [b'', b'/1', b' end'][int(rng.integers(0, 3))]
    out = []
    for i in range(n):
        f = []
        for k in kinds:
            if k == 'const': f.append(b'K9')
            elif k == 'small': f.append(b'%d' % int(rng.integers(0, 40)))
            elif k == 'big': f.append(b'%d' % int(rng.integers(0, 3000000000)))
            elif k == 'neg': f.append(b'%d' % int(rng.integers(-500, 500)))
            elif k == 'word': f.append(vocab[int(rng.integers(0, len(vocab)))])
            elif k == 'pad': f.append(b'%04d' % int(rng.integers(0, 3000)))
            else: f.append(vocab[int(rng.integers(0, 4))] if rng.random() < 0.3 else b'%d' % int(rng.integers(0, 9)))
        name = prefix
        for j, x in enumerate(f):
            name += x
            if j < nf - 1: name += seps[j].encode()
        out.append(name + suffix)
    return out


def test_random_grammars_differential(ctx):
    rng = np.random.default_rng(20261003 + HUNT)
    tally = {'ok': 0, 'error': 0, 'declined': 0}
    FUSED_TALLY.update(ok=0, declined=0)
    for case in range(120 * (1 + 9 * (HUNT > 0))):
        n = int(rng.integers(2, 400))
        tally[_check(ctx, _fastq(_random_family(rng, n)))] += 1
    assert tally['ok'] >= 40, tally           # the device path answers a solid share of random grammars


def test_fused_pass_random_decimal_grammars(ctx):
    """Random grammars whose fields are all plain decimals (what sequencers write): the fused pass answers nearly all of them --
    and every answer is the oracle's."""
    rng = np.random.default_rng(77 + HUNT)
    FUSED_TALLY.update(ok=0, declined=0)
    seps_pool = ':_/#=;, '
    for case in range(60 * (1 + 9 * (HUNT > 0))):
        n = int(rng.integers(2, 3000))
        nf = int(rng.integers(2, 7))
        seps = [seps_pool[int(rng.integers(0, len(seps_pool)))] for _ in range(nf - 1)]
        kinds = [['lane', 'tile', 'coord', 'serial', 'wide'][int(rng.integers(0, 5))] for _ in range(nf)]
        prefix = [b'@', b'@RUN7:', b'@M0123_45 '][int(rng.integers(0, 3))]
        suffix = [b'', b'/1', b' end'][int(rng.integers(0, 3))]
        period = int(rng.integers(1, 5))
        names = []
        for i in range(n):
            f = []
            for k in kinds:
                if k == 'lane': f.append(b'%d' % (1 + i % period))
                elif k == 'tile': f.append(b'%d' % (1101 + int(rng.integers(0, 64))))
                elif k == 'coord': f.append(b'%d' % int(rng.integers(1000, 30000)))
                elif k == 'serial': f.append(b'%d' % i)
                else: f.append(b'%d' % int(rng.integers(0, 999999999)))
            name = prefix
            for j, x in enumerate(f):
                name += x
                if j < nf - 1: name += seps[j].encode()
            names.append(name + suffix)
        _check(ctx, _fastq(names))
    assert FUSED_TALLY['ok'] >= 40, FUSED_TALLY


def test_fused_pass_demotion_and_offsets(ctx):
    """The fused pass on columns that exercise every typing branch it answers: mapping -> integers by the 10 000 / 20 000 / final
    checkpoints, small mappings that turn into integers with and without an offset, wide ranges judged on the checkpoints below
    2^21, and the cases it must hand back (a mapping whose values are too far apart, a value range beyond 2^20 that no
    checkpoint demotes)."""
    n = 23000
    names = [b'@q:%d:%d:%d:%d' % ((i % 1001) * 3, (i % 1000 if i <= 10000 else i % 2001) * 1000, 1101 + i % 64, i % 4) for i in range(n)]
    assert _check(ctx, _fastq(names), must_answer=True, fused_must_answer=True) == 'ok'
    n = 15000
    names = [b'@q:%d:%d' % (i % 7, (i % 1000) if i < 12000 else i) for i in range(n)]
    assert _check(ctx, _fastq(names), must_answer=True, fused_must_answer=True) == 'ok'
    names = [b'@r_%d_%d' % ([5, 70000, 12345678, 31][i % 4], i % 3) for i in range(400)]          # stays a mapping of strings
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok' and _fused(ctx, _fastq(names))[0] == 'declined'
    names = [b'@r_%d_%d' % ((i * 999331) % 3000000, i % 3) for i in range(30000)]                  # range 3 M > 2^20 slots
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
    names = [b'@r_%d_%d' % (i, 250 + i % 10) for i in range(70000)]                                # u4 / offset columns
    assert _check(ctx, _fastq(names), must_answer=True, fused_must_answer=True) == 'ok'
    # a first QNAME beyond 64 bytes (the guess kernel builds line 1's character table a lane per position up to 64, serially beyond)
    long_prefix = b'@' + b'instrument-0123456789-instrument-0123456789-instrument-0123456789-instrument.run'
    names = [long_prefix + b':%d:%d:%d' % (1 + i % 4, 1101 + i % 64, 1000 + (i * 7919) % 29000) for i in range(12000)]
    assert len(names[0]) > 64 and _check(ctx, _fastq(names), must_answer=True, fused_must_answer=True) == 'ok'
    # reads that break the layout late in the file: the fused pass must notice each of them
    base = [b'@r_%d_%d' % (i % 50, i) for i in range(20000)]
    for bad in (b'@r_7_x9', b'@r_7_007', b'@r_7', b'@r_7_8_9', b'@q_7_8', b'@r_7_+8', b'@r_7_', b'@r_7_12345678901'):
        names = list(base); names[17000] = bad
        want = _oracle(_fastq(names))
        _check_fused(ctx, _fastq(names), want)
        assert _fused(ctx, _fastq(names))[0] == 'declined', bad


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('UQ_TYPING_FUZZ_N', '10'))))      # UQ_TYPING_FUZZ_N=300 for a longer hunt
def test_fused_pass_typing_random(ctx, seed):
    """Column typing across the reference's checkpoints (10 000, 20 000, 40 000 reads and the last one: uq.py:586-602): random files of 9 000 -
    45 000 reads whose columns grow their sets of distinct values in different ways -- constant, cyclic with a random period, serial, random
    in a random range, cyclic up to a random read and serial after it, a few values with a late newcomer.  Whatever the fused pass answers
    is the oracle's answer (descriptions, dtypes, offsets, arrays); the exact device path likewise."""
    rng = np.random.default_rng(31_000 + seed)
    n = int(rng.integers(9000, 45000))
    nf = int(rng.integers(1, 5))
    kinds = []
    for _ in range(nf):
        k = ['const', 'cyc', 'serial', 'rand', 'late', 'newcomer'][int(rng.integers(0, 6))]
        kinds.append((k, int(rng.choice([2, 7, 64, 255, 256, 257, 900, 1100, 3000, 4095, 4097, 70000, 1 << 20, (1 << 20) + 5, 3_000_000])),
                      int(rng.integers(0, n)), int(rng.choice([0, 1, 1000, 65000, 10 ** 6]))))
    seps = [':_/#;'[int(rng.integers(0, 5))] for _ in range(nf)]
    cols = []
    for k, p, t0, base in kinds:
        i = np.arange(n, dtype=np.int64)
        if k == 'const': v = np.full(n, base)
        elif k == 'cyc': v = base + i % p
        elif k == 'serial': v = base + i
        elif k == 'rand': v = base + rng.integers(0, p, n)
        elif k == 'late': v = base + np.where(i < t0, i % min(p, 997), i)
        else: v = base + np.where(i == t0, p, i % 3)
        cols.append(v)
    names = []
    for r in range(n):
        name = b'@run'
        for c in range(nf):
            name += seps[c].encode() + b'%d' % int(cols[c][r])
        names.append(name)
    assert _check(ctx, _fastq(names)) in ('ok', 'error', 'declined')


def test_mutated_names_differential(ctx):
    """Adversarial for the closed form of uq.py:394-413: names are point mutations of line 1 over a tiny alphabet,
    so characters enter the separator table at different records and are knocked out before / after entering."""
    rng = np.random.default_rng(7 + HUNT)
    alphabet = np.frombuffer(b'1234:_/a', dtype=np.uint8)
    digits = np.frombuffer(b'1234', dtype=np.uint8)
    tally = {'ok': 0, 'error': 0, 'declined': 0}
    for case in range(600 * (1 + 9 * (HUNT > 0))):
        if case % 3 == 0:
            line1 = b'@' + bytes(rng.choice(alphabet, size=int(rng.integers(5, 14))))
        else:       # digit groups joined by separators: mostly encodable
            line1 = b'@' + b''.join(bytes(rng.choice(digits, size=int(rng.integers(1, 4)))) + bytes(rng.choice(alphabet[4:7], size=1))
                                    for _ in range(int(rng.integers(2, 5)))) + bytes(rng.choice(digits, size=2))
        names = [line1]
        for _ in range(int(rng.integers(1, 12))):
            q = bytearray(line1)
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(1, len(q)))
                op = rng.random()
                if case % 2:             # gentle: digits change, separators stay
                    if q[p] in digits: q[p] = int(rng.choice(digits))
                elif op < 0.7: q[p] = int(rng.choice(digits)) if (q[p] in digits and rng.random() < 0.9) else int(rng.choice(alphabet))
                elif op < 0.85: q.insert(p, int(rng.choice(alphabet)))
                elif len(q) > 3: del q[p]
            names.append(bytes(q))
        tally[_check(ctx, _fastq(names))] += 1
    assert tally['ok'] >= 20 and tally['error'] >= 20, tally


def test_cli_uses_device_qname_path(ctx, tmp_path):
    from uq_amd import uq
    fq = synth.fastq(99, 4000, 40)
    inp = tmp_path / 'in.fastq'; inp.write_bytes(fq)
    outs = {}
    for k, flag in enumerate(([], ['--exact-qname'], ['--host-qname'])):
        out = tmp_path / ('o%d.uQ' % k)
        args = uq.build_parser().parse_args(['-i', str(inp), '-o', str(out), '--quiet'] + flag)
        uq.validate_args(args)
        s = uq.Session(args, ctx=ctx)
        s.encode()
        outs[k] = (s.qname_path, O.read_tar(str(out)))
    assert [outs[k][0] for k in range(3)] == ['fused', 'device', 'host-native']
    for k in (1, 2):
        assert outs[0][1][0] == outs[k][1][0] and outs[0][1][1] == outs[k][1][1]
