"""GPU end-to-end parity: the drop-in CLI (uq_amd.uq) against (i) the .uQ files the reference itself wrote
(tests/golden/, byte for byte per member) and (ii) the oracle on seeded synthetic inputs over the whole
--sort x --raw x --pattern surface; then decode round trips."""
import io
import json
import os
import re
import tarfile

import numpy as np
import pytest

import uq_oracle as O
from uq_amd import synth, uq

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN = sorted(f[:-5] for f in os.listdir(GOLD) if f.endswith('.json'))
REFUSED = [n for n in GOLDEN if n.endswith('_refused')]        # the reference gives up on these inputs: so must the CLI, with an error of its own
WRITTEN = [n for n in GOLDEN if n not in REFUSED]


def _run_encode(ctx, tmp_path, fastq_bytes, flags):
    inp = tmp_path / 'in.fastq'
    inp.write_bytes(fastq_bytes)
    out = tmp_path / 'out.uQ'
    args = uq.build_parser().parse_args(['-i', str(inp), '-o', str(out), '--quiet'] + flags)
    uq.validate_args(args)
    s = uq.Session(args, ctx=ctx)
    s.encode()
    with tarfile.open(out) as t:
        names = t.getnames()
        members = {m.name: t.extractfile(m).read() for m in t.getmembers()}
    config = json.loads(members.pop('config.json').decode())
    return config, members, names, str(out)


def _run_decode(ctx, path):
    # both decoders, every time: rows -> text in one kernel (the default) and uq_unpack + uq_emit_fastq
    texts = []
    from uq_amd import ops
    for extra in ([], ['--two-pass-decode']):
        ops.scribble_lds(ctx, 0xC3C3C3C3 ^ len(texts))
        args = uq.build_parser().parse_args(['-i', path, '--decode', '--quiet'] + extra)
        uq.validate_args(args)
        buf = io.BytesIO()
        uq.Session(args, ctx=ctx).decode(out=buf)
        texts.append(buf.getvalue())
    assert texts[0] == texts[1]
    return texts[0]


def _records(fq):
    lines = fq.split(b'\n')
    return [b'\n'.join(lines[i:i + 4]) for i in range(0, len(lines) - 1, 4)]


def _oracle_flags(flags):
    def opt(k, n):
        if k in flags:
            i = flags.index(k); return flags[i + 1:i + 1 + n]
    sort = opt('--sort', 1); sort = None if (sort is None or sort[0] == 'None') else sort[0]
    raw = None
    if '--raw' in flags:
        raw = []
        for x in flags[flags.index('--raw') + 1:]:
            if x.startswith('--'): break
            raw.append(x)
    return dict(sort=sort, raw=raw, pattern=opt('--pattern', 2), notricks='--notricks' in flags, pad='--pad' in flags)


@pytest.mark.parametrize('name', REFUSED)
def test_cli_refuses_what_the_reference_refuses(ctx, tmp_path, name):
    from uq_amd import qname
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    assert meta['reference_refuses']
    with pytest.raises((uq.UqError, qname.QnameError)):
        _run_encode(ctx, tmp_path, fq, meta['flags'])
    assert not os.path.exists(tmp_path / 'out.uQ') or os.path.getsize(tmp_path / 'out.uQ') == 0


@pytest.mark.parametrize('name', WRITTEN)
def test_cli_matches_reference_output(ctx, tmp_path, name):
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    ref_cfg, ref_members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, meta['flags'])
    assert set(members) == set(ref_members)
    for k in ref_cfg:
        if k in ('sort', 'raw', 'pattern'): continue
        assert json.loads(json.dumps(cfg[k])) == ref_cfg[k], k
    assert cfg['pattern'] == ref_cfg['pattern'] and sorted(map(str, cfg['raw'])) == sorted(map(str, ref_cfg['raw']))
    if meta['stable_patch'] or 'sort' not in ' '.join(meta['flags']) or '--sort None' in ' '.join(meta['flags']):
        for k in ref_members:
            assert members[k] == ref_members[k], k
    else:
        # reference ran numpy's unstable argsort (Q17): the sorted-on table, every unique table and the
        # sorted-on key are order-free; the others agree as multisets inside each tie group = after decoding
        from test_oracle_golden import assert_equal_up_to_tie_order
        assert_equal_up_to_tie_order(cfg, members, ref_cfg, ref_members)          # exact members + multiset per tie group
        assert sorted(_records(O.decode(ref_cfg, ref_members).encode('latin-1'))) == sorted(_records(_run_decode(ctx, path)))
    # QNAME members come out in numeric order (Q6)
    q = [n for n in names if n.startswith('QNAME_')]
    assert q == sorted(q, key=lambda s: int(s.split('_')[1].split('.')[0]))
    # decode round trip: exact text when unsorted, same multiset of records when sorted
    text = _run_decode(ctx, path)
    if name in ('fixed_n_newcode', 'two_ntrick_bases'):
        return      # Q9: the reference's own decoder cannot decode a new N quality code either
    if ref_cfg['sort'] == [None]: assert text == fq
    else: assert sorted(_records(text)) == sorted(_records(fq))


@pytest.mark.parametrize('name', WRITTEN)
def test_decoder_reads_reference_written_files(ctx, name):
    """Existing .uQ files (written by the reference itself) decode on the device to the reads they were made from."""
    if name in ('fixed_n_newcode', 'two_ntrick_bases'):
        pytest.skip('Q9: a new N quality code is not decodable by the reference either')
    fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    text = _run_decode(ctx, os.path.join(GOLD, name + '.uQ'))
    ref_cfg, ref_members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
    assert text.decode('latin-1') == O.decode(ref_cfg, ref_members)              # same order as the reference's decoder semantics
    if ref_cfg['sort'] == [None]: assert text == fq
    else: assert sorted(_records(text)) == sorted(_records(fq))
    # ... and against what the reference's OWN decoder printed for this file (tests/golden/<name>.refdecode.fastq):
    # SEQ, '+' and QUAL lines position by position; the QNAME line too unless the json records the reference's Q6 defect
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    rd = meta['reference_decode']
    if rd['seq_qual_lines'] == 'unavailable': return
    ref_lines = open(os.path.join(GOLD, name + '.refdecode.fastq'), 'rb').read().split(b'\n')[:-1]
    lines = text.split(b'\n')[:-1]
    assert len(lines) == len(ref_lines) and all(lines[k::4] == ref_lines[k::4] for k in (1, 2, 3))
    if rd['qname_lines'] == 'equal to the input': assert lines[0::4] == ref_lines[0::4]


MIXES = [(s, r, p) for s in (None, 'DNA', 'QUAL', 'QNAME')
         for r in ([], ['DNA'], ['QUAL', 'QNAME'], ['DNA', 'QUAL', 'QNAME'])
         for p in (['0.1', '0.2'], ['1.1', '1.2'], ['2.1', '2.2'], ['3.1', '3.2'])]


@pytest.mark.parametrize('sort,raw,pattern', MIXES[::3], ids=lambda v: str(v))
def test_cli_matches_oracle_all_mixes(ctx, tmp_path, sort, raw, pattern):
    fq = synth.fastq(20261003 + 30, 1500, (30, 61), n_rate=2, dup='both', dup_templates=40)
    flags = (['--sort', sort] if sort else []) + (['--raw'] + raw if raw else []) + ['--pattern'] + pattern
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, flags)
    ocfg, omembers, _ = O.encode(fq, sort=sort, raw=raw or None, pattern=pattern)
    assert set(members) == set(omembers)
    for k in omembers:
        assert members[k] == omembers[k], k
    text = _run_decode(ctx, path)
    assert text.decode('latin-1') == O.decode(ocfg, omembers)
    if sort is None: assert text == fq
    else: assert sorted(_records(text)) == sorted(_records(fq))


def test_cli_config1_fixed_100bp(ctx, tmp_path):
    """BASELINE.json configs[0]: 10k x 100bp, --sort None --raw DNA QUAL QNAME --pattern 0.1 0.1."""
    fq = open(os.path.join(GOLD, 'cfg1_10k_100bp.fastq'), 'rb').read()
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--sort', 'None', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '0.1', '0.1'])
    assert cfg['bits_per_base'] == 2 and cfg['bits_per_quality'] == 6 and cfg['reads'] == 10000
    assert np.load(io.BytesIO(members['DNA.raw'])).shape == (10000, 25)
    assert np.load(io.BytesIO(members['QUAL.raw'])).shape == (10000, 75)
    assert _run_decode(ctx, path) == fq


def test_cli_test_mode_without_compressor(ctx, tmp_path):
    """--test without --compressor forces sort None and pattern 0.1 (Q3) and picks the smallest raw set."""
    fq = synth.fastq(20261003 + 31, 800, 40, dup='both', dup_templates=10)
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--test'])
    assert cfg['sort'] == [None] and cfg['pattern'] == ['0.1', '0.1']
    assert _run_decode(ctx, path) == fq


def test_cli_test_mode_with_compressor(ctx, tmp_path):
    """--test --compressor: the 8-layout sweep per table and the (raw set x sort) grid run on the device
    kernels; whatever wins must still decode to the input (as a multiset when a sort won)."""
    import shutil
    if shutil.which('gzip') is None: pytest.skip('no gzip')
    fq = synth.fastq(20261003 + 32, 600, 36, dup='both', dup_templates=8)
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--test', '--compressor', 'gzip -1', '--raw', 'DNA', 'QNAME'])
    assert cfg['pattern'][0] in uq.PATTERNS and cfg['pattern'][1] in uq.PATTERNS
    text = _run_decode(ctx, path)
    assert sorted(_records(text)) == sorted(_records(fq))
    # and the oracle builds the same members for the mix the sweep chose
    ocfg, omembers, _ = O.encode(fq, sort=cfg['sort'] if isinstance(cfg['sort'], str) else None, raw=[r for r in cfg['raw'] if r], pattern=cfg['pattern'])
    assert set(members) == set(omembers) and all(members[k] == omembers[k] for k in omembers)


def test_cli_edge_inputs(ctx, tmp_path):
    # one record; two records with the minimum of structure; a file without the final newline (Q21: refused)
    for fq in (b'@a:1:7\nACGTN\n+\nIHIH#\n@a:2:9\nACGTA\n+\nHIHII\n', b'@q:1:5\nA\n+\nI\n@q:2:6\nA\n+\nH\n'):
        cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--raw', 'DNA', 'QUAL', 'QNAME'])
        ocfg, omembers, _ = O.encode(fq, raw=['DNA', 'QUAL', 'QNAME'])
        assert all(members[k] == omembers[k] for k in omembers)
        assert _run_decode(ctx, path) == fq
    p = tmp_path / 'nonl.fastq'
    p.write_bytes(b'@q:1:5\nA\n+\nI\n@q:2:6\nC\n+\nI')
    with pytest.raises(uq.UqError):
        uq.Session(uq.validate_args(uq.build_parser().parse_args(['-i', str(p), '--quiet'])), ctx=ctx).encode()
    # reads longer than one LDS tile take the HBM-direct kernels (see test_cli_long_reads)
    L = 30000
    fq = b'@q:1:5\n' + b'A' * L + b'\n+\n' + b'I' * L + b'\n@q:2:6\n' + b'C' * L + b'\n+\n' + b'H' * L + b'\n'
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, [])
    assert _run_decode(ctx, path) == fq


def test_device_text_emit_matches_oracle_decode(ctx, tmp_path):
    """uq_emit_fastq: mapping columns (string table), integer columns with and without offset, negative
    numbers, a suffix, variable lengths -- the decoded text equals the oracle's decode and the input."""
    import random
    rnd = random.Random(11)
    recs = []
    for i in range(3000):
        L = rnd.randint(5, 60)
        seq = ''.join(rnd.choice('ACGT') for _ in range(L)); q = ''.join(rnd.choice('#5AI') for _ in range(L))
        recs.append('@run7 %s:%d:%d:%d/%s\n%s\n+\n%s\n' % (rnd.choice(['ab', 'c', 'Zed']), 70000 + i * 3, rnd.randint(-50, 50), i % 7, 'x1', seq, q))
    fq = ''.join(recs).encode()
    for flags in (['--raw', 'DNA', 'QUAL', 'QNAME'], ['--sort', 'None']):
        cfg, members, names, path = _run_encode(ctx, tmp_path, fq, flags)
        kinds = [(c['format'], c.get('offset')) for c in cfg['QNAME_columns']]
        assert ('mapping', None) in kinds and ('integers', True) in kinds and ('integers', False) in kinds
        ocfg, omembers, _ = O.encode(fq, raw=[f for f in flags[1:] if f not in ('None',)] if flags[0] == '--raw' else None)
        assert all(members[k] == omembers[k] for k in omembers)
        text = _run_decode(ctx, path)
        assert text.decode('latin-1') == O.decode(ocfg, omembers)
        assert text == fq


def test_cli_errors(ctx, tmp_path):
    p = tmp_path / 'bad.fastq'
    p.write_bytes(b'@a:1\nACGT\n+\nIIII\n@a:2\nAC\n+\nIII\n')
    args = uq.build_parser().parse_args(['-i', str(p), '--quiet'])
    with pytest.raises(uq.UqError):
        uq.Session(uq.validate_args(args), ctx=ctx).encode()
    p.write_bytes(b'@a:1\nACGT\n+\nIIII\n@a:2\nAC\n')
    with pytest.raises(uq.UqError):
        uq.Session(uq.validate_args(uq.build_parser().parse_args(['-i', str(p), '--quiet'])), ctx=ctx).encode()
    with pytest.raises(uq.UqError):
        uq.validate_args(uq.build_parser().parse_args(['-i', str(p), '--pattern', '0.1']))
    with pytest.raises(uq.UqError):
        uq.validate_args(uq.build_parser().parse_args(['-i', str(p), '--sort', 'bogus']))


def _fuzz_case(rng):
    """A random FASTQ (alphabets, quality ranges, lengths, QNAME family) and random CLI flags."""
    scale = int(os.environ.get('UQ_FUZZ_SCALE', '1'))          # > 1: many tiles per kernel, reads up to 508 bp (slow oracle)
    n = int(rng.integers(1, 500)) * scale
    bases = [b'ACGT', b'ACGTN', b'ACGTNRYKM', b'acgtn', b'AC', b'ACGTUWSBDHVN-.'][int(rng.integers(0, 6))]
    quals = [bytes(range(33, 74)), b'#-<F', bytes(range(64, 105)), b'!I', bytes(range(35, 127, 3)), b'5'][int(rng.integers(0, 6))]
    fixed = rng.random() < 0.4
    lo = int(rng.integers(1, 40)); hi = lo if fixed else lo + int(rng.integers(1, 120 if scale == 1 else 460))
    if os.environ.get('UQ_FUZZ_LONG'):                          # few reads, up to 9 kbp: tiles packed piece by piece, reads beyond a tile (the exact kernels)
        n = int(rng.integers(1, 60))
        lo = int(rng.integers(1, 3000)); hi = lo if fixed else lo + int(rng.integers(1, 6000))
    n_single_quality = rng.random() < 0.5            # N always with one quality (the N-trick applies)
    fam = int(rng.integers(0, 8))                    # 0-5: separators the reference copes with; 6, 7: families it tends to refuse
    s1, s2 = b':_#;='[int(rng.integers(0, 5))], b':_#;='[int(rng.integers(0, 5))]
    recs = []
    B = np.frombuffer(bases, np.uint8); Q = np.frombuffer(quals, np.uint8)
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = rng.choice(B, L); q = rng.choice(Q, L)
        if n_single_quality and (b'N' in bases):
            q[s == ord('N')] = Q[0]
        if fam <= 5: name = b'@q%c%d%c%d' % (s1, i % 7, s2, 3 * i + 1)
        elif fam == 6: name = b'@run7_%d/%d' % (1000 - i, 1 + i % 2)
        else: name = b'@x.%s.%d' % ([b'aa', b'b', b'cde'][i % 3], i)
        recs.append(name + b'\n' + bytes(s) + b'\n+\n' + bytes(q) + b'\n')
    flags = []
    if rng.random() < 0.6: flags += ['--sort', ['DNA', 'QUAL', 'QNAME', 'None'][int(rng.integers(0, 4))]]
    raw = [t for t in ('DNA', 'QUAL', 'QNAME') if rng.random() < 0.5]
    if raw: flags += ['--raw'] + raw
    if rng.random() < 0.5:
        ids = ['0.1', '0.2', '1.1', '1.2', '2.1', '2.2', '3.1', '3.2']
        flags += ['--pattern', ids[int(rng.integers(0, 8))], ids[int(rng.integers(0, 8))]]
    if rng.random() < 0.3: flags.append('--notricks')
    if rng.random() < 0.3: flags.append('--pad')
    return b''.join(recs), flags


@pytest.mark.parametrize('seed', range(int(os.environ.get('UQ_FUZZ_FROM', '0')), int(os.environ.get('UQ_FUZZ_FROM', '0')) + int(os.environ.get('UQ_FUZZ_N', '48'))))      # UQ_FUZZ_N=2000 for a longer hunt, UQ_FUZZ_FROM=25000 for seeds not hunted before
def test_cli_fuzz_against_oracle(ctx, tmp_path, seed):
    """Differential fuzz of the whole CLI: random alphabets / bit widths / lengths / QNAME families / flag mixes."""
    fq, flags = _fuzz_case(np.random.default_rng(1000 + seed))
    of = _oracle_flags(flags)
    try:
        ocfg, omembers, _ = O.encode(fq, **of)
    except (O.UqError, ValueError, IndexError, re.error):   # the reference refuses this input (e.g. no QNAME separator): so must the
        from uq_amd import qname                             # CLI, with ITS error types -- a crash of the product is not a refusal
        with pytest.raises((uq.UqError, qname.QnameError)):
            _run_encode(ctx, tmp_path, fq, flags)
        return
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, flags + os.environ.get('UQ_FUZZ_FLAGS', '').split())   # e.g. --multi-pass
    assert set(members) == set(omembers)
    for k in omembers:
        assert members[k] == omembers[k], (k, flags)
    for k in ocfg:
        if k in ('sort', 'raw', 'pattern'): continue
        assert json.loads(json.dumps(cfg[k])) == json.loads(json.dumps(ocfg[k])), k
    if ocfg['N_qual'] and max(ocfg['N_qual'].values()) >= len(ocfg['qualities']):
        return                                  # Q9: a NEW N quality code is not decodable by the reference's own decoder either
    text = _run_decode(ctx, path)
    assert text.decode('latin-1') == O.decode(ocfg, omembers)
    if of['sort'] is None: assert text == fq


@pytest.mark.parametrize('shape', ['few-long-among-short', 'one-beyond-a-tile', 'all-long'])
def test_cli_mixed_read_lengths(ctx, tmp_path, shape):
    """Files whose longest read is far above the average: the pack kernel sizes its tiles from the average record and its rows from the longest one
    (a file of short reads with a few of 5 - 8 kbp once asked for more LDS than a CU has and the encode ended in an error); a read beyond one
    tile (30 kbp) takes the exact thread-per-read kernels.  Members == oracle, decode == input."""
    rng = np.random.default_rng(len(shape))
    if shape == 'few-long-among-short': lens = [int(x) for x in np.where(rng.random(1500) < 0.02, rng.integers(5000, 7800, 1500), rng.integers(30, 200, 1500))]
    elif shape == 'one-beyond-a-tile': lens = [int(x) for x in rng.integers(50, 150, 400)] + [30_000] + [int(x) for x in rng.integers(50, 150, 100)]
    else: lens = [int(x) for x in rng.integers(4000, 7900, 120)]
    recs = []
    for i, L in enumerate(lens):
        recs.append(b'@r:%d:%d\n' % (i % 5, i) + bytes(rng.choice(np.frombuffer(b'ACGT', np.uint8), L)) + b'\n+\n' + bytes(rng.integers(40, 75, L).astype(np.uint8)) + b'\n')
    fq = b''.join(recs)
    for flags in ([], ['--sort', 'DNA', '--raw', 'QUAL']):
        ocfg, omembers, _ = O.encode(fq, **_oracle_flags(flags))
        cfg, members, names, path = _run_encode(ctx, tmp_path, fq, flags)
        assert set(members) == set(omembers)
        for k in omembers: assert members[k] == omembers[k], (k, flags)
        text = _run_decode(ctx, path)
        assert text.decode('latin-1') == O.decode(ocfg, omembers)
        if not flags: assert text == fq


def test_cli_speculative_pack(ctx, tmp_path):
    """The encoder packs speculatively with decisions guessed from the head of the file itself, in the kernel that also counts the
    statistics (uq_pack_stats: two reads of the stream); a wrong guess falls back to the separate pack; --multi-pass never
    speculates.  `pack_path` says whether the speculative tables were kept.  The output never depends on the path taken."""
    a = synth.fastq(20261003 + 50, 3000, 60)
    b = synth.fastq(20261003 + 51, 2500, 60)                                   # same alphabets and length as a: a's decisions hold
    c = synth.fastq(20261003 + 52, 2000, (30, 61), n_rate=2)                   # variable length, N: they do not
    # d: the head (first 8192 reads) is ACGT only, an N turns up later: the head's guess fails, the fallback packs
    d = synth.fastq(20261003 + 53, 70_000, 40)
    k = d.rfind(b'\n', 0, len(d) - 200)
    k = d.rfind(b'\n', 0, d.rfind(b'\n+\n', 0, k))                             # start of a SEQ line near the end
    d = d[:k + 1] + b'N' + d[k + 2:]
    want = {}
    for mode, expect in (([], ['speculative', 'speculative', 'speculative', 'speculative', 'plain']), (['--multi-pass'], ['plain'] * 5)):
        uq.Session.last_params = None
        paths = []
        for i, fq in enumerate((a, b, c, b, d)):
            inp = tmp_path / ('in%d.fastq' % i); inp.write_bytes(fq)
            out = tmp_path / ('out%d.uQ' % i)
            args = uq.validate_args(uq.build_parser().parse_args(['-i', str(inp), '-o', str(out), '--quiet', '--raw', 'DNA', 'QUAL', 'QNAME'] + mode))
            s = uq.Session(args, ctx=ctx)
            s.encode()
            paths.append(s.pack_path)
            cfg, members = O.read_tar(str(out))
            if i not in want: want[i] = O.encode(fq, raw=['DNA', 'QUAL', 'QNAME'])[1]
            assert all(members[k] == want[i][k] for k in want[i]), (mode, i)
        assert paths == expect, (mode, paths)
    uq.Session.last_params = None


@pytest.mark.parametrize('flags', [[], ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME'], ['--notricks', '--pattern', '2.2', '1.1']],
                         ids=lambda v: '_'.join(v) or 'default')
def test_cli_long_reads(ctx, tmp_path, flags):
    """Reads far beyond one LDS tile (long-read platforms): pack / unpack / emit take their HBM-direct paths."""
    rng = np.random.default_rng(31)
    recs = []
    for i, L in enumerate([12000, 30000, 70000, 100, 25000, 60001, 7]):
        seq = rng.choice(np.frombuffer(b'ACGT', np.uint8), L)
        qual = rng.integers(35, 75, L, dtype=np.uint8)
        n_at = rng.random(L) < 0.01
        seq[n_at] = ord('N'); qual[n_at] = ord('!')
        recs.append(b'@ont:%d:%d\n' % (i, 1000 + 7 * i) + bytes(seq) + b'\n+\n' + bytes(qual) + b'\n')
    fq = b''.join(recs)
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, flags)
    ocfg, omembers, _ = O.encode(fq, **_oracle_flags(flags))
    assert set(members) == set(omembers)
    for k in omembers:
        assert members[k] == omembers[k], k
    text = _run_decode(ctx, path)
    assert text.decode('latin-1') == O.decode(ocfg, omembers)
    if '--sort' not in flags: assert text == fq


def test_decode_many_qname_columns(ctx, tmp_path):
    """Twelve QNAME fields: a tile holds more (record, field) items than the workgroup has lanes, so the emit / decode
    kernels take their looped field passes; mixed widths, a mapping column and negative numbers included."""
    import random
    rnd = random.Random(5)
    recs = []
    for i in range(2500):
        L = rnd.randint(20, 90)
        seq = ''.join(rnd.choice('ACGTN') for _ in range(L)); q = ''.join(rnd.choice('#+5AFI') for _ in range(L))
        f = [i, rnd.randint(0, 9), rnd.randint(-5, 5), 70000 + rnd.randint(0, 10 ** 6), rnd.choice(['x', 'yy', 'zzz']), rnd.randint(0, 255),
             rnd.randint(0, 65535), 2 ** 33 + rnd.randint(0, 1000), i % 3, rnd.randint(100, 999), i * 7, rnd.randint(0, 1)]
        recs.append('@m%d:%d_%d %d/%s;%d,%d:%d:%d:%d:%d:%d#z\n%s\n+\n%s\n' % (*f, seq, q))
    fq = ''.join(recs).encode()
    for flags in (['--raw', 'DNA', 'QUAL', 'QNAME'], []):
        cfg, members, names, path = _run_encode(ctx, tmp_path, fq, flags)
        assert len(cfg['QNAME_columns']) >= 10
        assert sorted(_records(_run_decode(ctx, path))) == sorted(_records(fq))
        if flags: assert _run_decode(ctx, path) == fq


@pytest.mark.parametrize('shape', ['long-strings', 'twelve-columns', 'two-columns', 'long-reads'])
def test_decode_variable_lengths_straight_to_hbm(ctx, tmp_path, shape):
    """Variable read lengths with the lookup-free alphabet (2-bit bases, one contiguous quality range): uq_decode_fastq writes
    SEQ / QUAL groups straight to HBM and stages only the QNAME lines (decode_stream_kernel).  Shapes: QNAME strings far beyond
    the staging share of a line (a wave per line instead), more (record, field) items than lanes, lines
    shorter than one group, reads of a few thousand bases (few reads per tile, flat group loop)."""
    import random
    rnd = random.Random(11)
    recs = []
    n = 300 if shape == 'long-reads' else 6000
    for i in range(n):
        L = rnd.choice([1, 2, 7, 8, 9, 15, 16, 17, 40, 63, 64, 65, 150]) if shape != 'long-reads' else rnd.choice([5, 800, 2047, 2048, 4100])
        seq = ''.join(rnd.choice('ACGT') for _ in range(L)); q = ''.join(chr(rnd.randint(40, 70)) for _ in range(L))
        if shape == 'long-strings': name = '@%s:%d:%s' % (rnd.choice(['flowcell-' + 'x' * 70, 'lane-' + 'y' * 130, 'z']), i, rnd.choice(['a' * 40, 'b']))
        elif shape == 'twelve-columns': name = '@m%d:%d_%d %d/%d;%d,%d:%d:%d:%d:%d:%d#z' % (i, rnd.randint(0, 9), rnd.randint(-5, 5), 70000 + rnd.randint(0, 10 ** 6), rnd.randint(0, 3),
                                                                                     rnd.randint(0, 255), rnd.randint(0, 65535), 2 ** 33 + rnd.randint(0, 1000), i % 3, rnd.randint(100, 999), i * 7, rnd.randint(0, 1))
        else: name = '@r.%d %d' % (i, rnd.randint(0, 10 ** 9))
        recs.append('%s\n%s\n+\n%s\n' % (name, seq, q))
    fq = ''.join(recs).encode()
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--raw', 'DNA', 'QUAL', 'QNAME'])
    assert cfg['bits_per_base'] == 2 and cfg['variable_read_lengths']
    assert _run_decode(ctx, path) == fq


def _rewrite_tar(src, dst, edit):
    """copy the container `src` to `dst` with member payloads changed by edit(name, bytes) -> bytes"""
    with tarfile.open(src) as t, tarfile.open(dst, 'w') as o:
        for m in t.getmembers():
            data = edit(m.name, t.extractfile(m).read())
            ti = tarfile.TarInfo(m.name); ti.size = len(data)
            o.addfile(ti, io.BytesIO(data))


def _npy_set(data, index, value):
    """the .npy member `data` with element `index` of its flat payload set to `value`"""
    f = io.BytesIO(data)
    np.lib.format.read_magic(f)
    shape, fortran, dtype = np.lib.format.read_array_header_1_0(f)
    hdr = f.tell()
    a = np.frombuffer(data[hdr:], dtype=dtype).copy()
    a[index] = value
    return data[:hdr] + a.tobytes()


def test_decoder_refuses_damaged_containers(ctx, tmp_path):
    """A .uQ file is untrusted input.  A key that points beyond its table, a mapping code beyond its string table and a
    truncated member each end in a UqError (the reference's numpy raises IndexError / ValueError on them, uq.py:944-973,
    1016) -- not in an out-of-bounds device read, and not in plausible but wrong FASTQ."""
    import random
    rnd = random.Random(5)
    seqs = [''.join(rnd.choice('ACGT') for _ in range(30)) for _ in range(5)]
    recs = []
    for i in range(400):
        recs.append('@run %s:%d\n%s\n+\n%s\n' % (rnd.choice(['ab', 'c', 'Zed']), i, rnd.choice(seqs), ''.join(rnd.choice('#5AI') for _ in range(30))))
    fq = ''.join(recs).encode()
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--raw', 'QUAL', 'QNAME'])      # DNA keyed: 5 distinct rows, u1 key
    assert 'DNA.key' in members and cfg['QNAME_columns'][0]['format'] == 'mapping' and len(cfg['QNAME_columns'][0]['map']) == 3
    assert _run_decode(ctx, path) == fq

    def decode(p):
        args = uq.build_parser().parse_args(['-i', str(p), '--decode', '--quiet'])
        uq.validate_args(args)
        buf = io.BytesIO()
        uq.Session(args, ctx=ctx).decode(out=buf)
        return buf.getvalue()

    bad = tmp_path / 'bad.uQ'
    _rewrite_tar(path, bad, lambda n, d: _npy_set(d, 123, 5) if n == 'DNA.key' else d)            # == nunique
    with pytest.raises(uq.UqError, match='DNA.key.*entry 123'):
        decode(bad)
    _rewrite_tar(path, bad, lambda n, d: _npy_set(d, 77, 3) if n == 'QNAME_1.raw' else d)        # == len(map)
    with pytest.raises(uq.UqError, match='QNAME column 1.*entry 77'):
        decode(bad)
    _rewrite_tar(path, bad, lambda n, d: d[:-100] if n == 'QUAL.raw' else d)                      # truncated member
    with pytest.raises(uq.UqError, match='QUAL.raw.*damaged'):
        decode(bad)
    _rewrite_tar(path, bad, lambda n, d: d[:-1] if n == 'QNAME_2.raw' else d)
    with pytest.raises(uq.UqError, match='QNAME_2.raw.*damaged'):
        decode(bad)
    _rewrite_tar(path, bad, lambda n, d: d)                                                       # the copy itself decodes
    assert decode(bad) == fq


def test_decode_into_appending_file(ctx, tmp_path):
    """`uq --decode >> out.fastq`: pwrite() ignores its offset on O_APPEND descriptors, so the parallel chunk writer must not
    be used there (uq_amd/hostio.py)."""
    from uq_amd import hostio
    fq = synth.fastq(20261003 + 62, 4000, 100)
    cfg, members, names, path = _run_encode(ctx, tmp_path, fq, ['--raw', 'DNA', 'QUAL', 'QNAME'])
    out = tmp_path / 'appended.fastq'
    out.write_bytes(b'@head\nA\n+\nI\n')
    old = hostio.CHUNK
    args = uq.build_parser().parse_args(['-i', path, '--decode', '--quiet'])
    uq.validate_args(args)
    s = uq.Session(args, ctx=ctx)
    s.io.chunk = 64 << 10                       # many chunks in flight: completion order != file order on the parallel path
    s.io.nbuf = 8
    with open(out, 'ab') as fh:
        s.decode(out=fh)
    assert out.read_bytes() == b'@head\nA\n+\nI\n' + fq
