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
    return b''.join(n + b'\nACGT\n+\nIIII\n' for n in names)


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


def _check(ctx, fq, must_answer=False):
    want, got = _oracle(fq), _device(ctx, fq)
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
        assert _check(ctx, fq, must_answer=True) == 'ok'
    for n in (1, 2, 3, 11):
        _check(ctx, synth.fastq(7, n, 20), must_answer=True)      # tiny files: same answer or same refusal


def test_golden_fastq(ctx):
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    for f in sorted(os.listdir(gold)):
        if f.endswith('.fastq'):
            assert _check(ctx, open(os.path.join(gold, f), 'rb').read(), must_answer=True) == 'ok'


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
    assert _check(ctx, _fastq(names), must_answer=True) == 'ok'
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
    rng = np.random.default_rng(20261003)
    tally = {'ok': 0, 'error': 0, 'declined': 0}
    for case in range(120):
        n = int(rng.integers(2, 400))
        tally[_check(ctx, _fastq(_random_family(rng, n)))] += 1
    assert tally['ok'] >= 40, tally           # the device path answers a solid share of random grammars


def test_mutated_names_differential(ctx):
    """Adversarial for the closed form of uq.py:394-413: names are point mutations of line 1 over a tiny alphabet,
    so characters enter the separator table at different records and are knocked out before / after entering."""
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b'1234:_/a', dtype=np.uint8)
    digits = np.frombuffer(b'1234', dtype=np.uint8)
    tally = {'ok': 0, 'error': 0, 'declined': 0}
    for case in range(600):
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
    for flag in ([], ['--host-qname']):
        out = tmp_path / ('o%d.uQ' % len(flag))
        args = uq.build_parser().parse_args(['-i', str(inp), '-o', str(out), '--quiet'] + flag)
        uq.validate_args(args)
        s = uq.Session(args, ctx=ctx)
        s.encode()
        outs[len(flag)] = (s.qname_path, O.read_tar(str(out)))
    assert outs[0][0] == 'device' and outs[1][0] == 'host-native'
    assert outs[0][1][0] == outs[1][1][0] and outs[0][1][1] == outs[1][1][1]
