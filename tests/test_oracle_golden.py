"""CPU: the oracle (oracle/uq_oracle.py) against the outputs of the reference itself (tests/golden/*.uQ,
written by tests/golden/make_golden.py) -- byte for byte per tar member -- plus the README's worked
geometry and the hand-checked micro-vectors of SURVEY.md A.7.  This is what pins the oracle."""
import glob
import io
import json
import os

import numpy as np
import pytest

import uq_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN = sorted(os.path.basename(f)[:-5] for f in glob.glob(os.path.join(GOLD, '*.json')))
REFUSED = [n for n in GOLDEN if n.endswith('_refused')]        # inputs the reference itself gives up on: the fixture is the refusal
WRITTEN = [n for n in GOLDEN if n not in REFUSED]
Q9 = ['fixed_n_newcode', 'two_ntrick_bases']                   # a NEW N quality code (uq.py:493-494): no decoder can read these files back, the reference's included


def flags_to_kwargs(flags):
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


def test_golden_set_is_complete():
    assert len(WRITTEN) >= 40 and len(REFUSED) >= 6
    for name in GOLDEN:
        for ext in ('.fastq', '.json') + (() if name in REFUSED else ('.uQ',)):
            assert os.path.exists(os.path.join(GOLD, name + ext))
    for name in REFUSED:
        meta = json.load(open(os.path.join(GOLD, name + '.json')))
        assert meta['reference_refuses'] and meta['reference_says'] and not os.path.exists(os.path.join(GOLD, name + '.uQ'))


def test_golden_set_covers_more_than_one_qname_grammar():
    """the fixtures of rounds 1-3 all shared ':' separators and four small integer columns.  The set must hold mapping
    columns, a suffix, other separators (a space among them), uint32 / uint64 columns, offsets, and more than one column count."""
    seps, fmts, dtypes, suffixes, offsets, ncols = set(), set(), set(), set(), set(), set()
    for name in WRITTEN:
        cfg, _ = O.read_tar(os.path.join(GOLD, name + '.uQ'))
        seps |= set(cfg['QNAME_separators']); suffixes.add(cfg['QNAME_suffix']); ncols.add(len(cfg['QNAME_columns']))
        for c in cfg['QNAME_columns']:
            fmts.add(c['format']); dtypes.add(c['dtype']); offsets.add(c.get('offset'))
    assert seps >= set(': _-=;') and fmts == {'integers', 'mapping'} and dtypes >= {'uint8', 'uint16', 'uint32', 'uint64'}
    assert len(suffixes) >= 3 and {True, False} <= offsets and len(ncols) >= 4


def test_golden_set_covers_the_width_ladder():
    """... and the alphabets: every DNA width from 2 to 4 bits and every QUAL width from 2 to 8 occurs in a reference-written container (uq.py:497-503,
    534-540), with and without N-trick bases, two of them at once included (Q11: order of first appearance; Q9: a new quality code)."""
    bd, bq, ntrick = set(), set(), set()
    for name in WRITTEN:
        cfg, _ = O.read_tar(os.path.join(GOLD, name + '.uQ'))
        bd.add(cfg['bits_per_base']); bq.add(cfg['bits_per_quality']); ntrick.add(len(cfg['N_qual']))
    assert bd >= {2, 3, 4} and bq >= {2, 4, 5, 6, 7, 8} and ntrick >= {0, 1, 2}, (bd, bq, ntrick)


@pytest.mark.parametrize('name', REFUSED)
def test_oracle_refuses_what_the_reference_refuses(name):
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    with pytest.raises(O.UqError) as e:
        O.encode(fq, **flags_to_kwargs(meta['flags']))
    # where the reference says why in its own words, the restatement says the same
    if not meta['reference_says'].startswith('re.error'):
        assert meta['reference_says'][:40] in str(e.value) or str(e.value)[:40] in meta['reference_says'] or 'Sorry' in str(e.value)


@pytest.mark.parametrize('name', WRITTEN)
def test_oracle_matches_reference_members(name):
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    ref_cfg, ref_members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
    cfg, members, tables = O.encode(fq, **flags_to_kwargs(meta['flags']))
    assert set(members) == set(ref_members)
    for k in ref_cfg:
        if k in ('sort', 'raw', 'pattern'): continue
        assert json.loads(json.dumps(cfg[k])) == ref_cfg[k], k
    unstable = (not meta['stable_patch']) and cfg['sort'] not in ([None], None)
    if not unstable:
        for k in ref_members:
            assert members[k] == ref_members[k], k
    else:
        assert_equal_up_to_tie_order(cfg, members, ref_cfg, ref_members)


def records_in_stored_order(cfg, members):
    """[(QNAME line, SEQ, QUAL)] in the order the container stores the reads (decoded by the oracle's decoder, which
    reads QNAME columns by number -- the reference's own decoder permutes them, Q6)."""
    lines = O.decode(cfg, members).split('\n')[:-1]
    return list(zip(lines[0::4], lines[1::4], lines[3::4]))


def assert_equal_up_to_tie_order(cfg, members, ref_cfg, ref_members):
    """The Q17 parity rule against the UNPATCHED reference (numpy's default argsort is unstable, uq.py:775, 796, 816, 833):
      * exact: every unique table, every raw table / key OF THE SORTED-ON table;
      * position by position the sorted-on field is the same, and inside every run of equal sorted-on values (a tie
        group) both files hold the same multiset of whole records -- only the order inside a group may differ;
      * a member that is the same bytes needs no argument."""
    on = cfg['sort']
    exact = {'DNA': ('DNA', 'DNA.key', 'DNA.raw'), 'QUAL': ('QUAL', 'QUAL.key', 'QUAL.raw'),
             'QNAME': tuple(k for k in ref_members if k.startswith('QNAME'))}[on]
    for k in ref_members:
        unique_table = k in ('DNA', 'QUAL') or (k.startswith('QNAME_') and not k.endswith('.raw'))
        if k in exact or unique_table:
            assert members[k] == ref_members[k], k
    ours, theirs = records_in_stored_order(cfg, members), records_in_stored_order(ref_cfg, ref_members)
    field = {'QNAME': 0, 'DNA': 1, 'QUAL': 2}[on]
    assert [r[field] for r in ours] == [r[field] for r in theirs], 'sorted-on field differs'
    assert len(ours) == len(theirs)
    i, groups, moved = 0, 0, 0
    while i < len(ours):
        j = i
        while j < len(ours) and ours[j][field] == ours[i][field]: j += 1
        assert sorted(ours[i:j]) == sorted(theirs[i:j]), 'tie group at stored position %d holds different records' % i
        groups += j - i > 1
        moved += ours[i:j] != theirs[i:j]
        i = j
    return groups, moved


def test_tie_rule_is_not_vacuous():
    """The unpatched sorted fixtures do contain tie groups, and for at least one of them the reference's unstable order
    differs from file order -- so the multiset rule is exercised, not just satisfied by identical bytes."""
    seen_groups = seen_moved = 0
    for name in WRITTEN:
        meta = json.load(open(os.path.join(GOLD, name + '.json')))
        if meta['stable_patch'] or '--sort' not in meta['flags'] or meta['flags'][meta['flags'].index('--sort') + 1] == 'None': continue
        fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
        ref_cfg, ref_members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
        cfg, members, _ = O.encode(fq, **flags_to_kwargs(meta['flags']))
        g, m = assert_equal_up_to_tie_order(cfg, members, ref_cfg, ref_members)
        seen_groups += g; seen_moved += m
    assert seen_groups > 20 and seen_moved > 0, (seen_groups, seen_moved)


@pytest.mark.parametrize('name', WRITTEN)
def test_reference_decoder_output_pins_seq_and_qual(name):
    """What the reference's OWN decoder printed for the file the reference wrote (<name>.refdecode.fastq): lines 2 and 4 of
    every record equal the oracle decoder's, position by position.  Line 1 is excluded where the json says why (Q6: the
    reference decoder permutes QNAME columns by tar member order)."""
    meta = json.load(open(os.path.join(GOLD, name + '.json')))
    rd = meta['reference_decode']
    path = os.path.join(GOLD, name + '.refdecode.fastq')
    if rd['seq_qual_lines'] == 'unavailable':
        assert not os.path.exists(path) and rd['why']
        pytest.skip('the reference decoder stops on this file: ' + rd['why'])
    ref_lines = open(path, 'rb').read().decode('latin-1').split('\n')[:-1]
    ref_cfg, ref_members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
    lines = O.decode(ref_cfg, ref_members).split('\n')[:-1]
    assert len(lines) == len(ref_lines)
    assert lines[1::4] == ref_lines[1::4] and lines[3::4] == ref_lines[3::4] and lines[2::4] == ref_lines[2::4]
    assert rd['seq_qual_lines'].startswith('equal to the input')
    if rd['qname_lines'] == 'equal to the input': assert lines[0::4] == ref_lines[0::4]
    else: assert 'Q6' in rd['why']


@pytest.mark.skipif(not os.path.exists('/root/reference/uq.py'), reason='the reference lives in the build container only')
def test_fixtures_regenerate_from_the_reference(tmp_path):
    """Fixture drift guard: run tests/golden/make_golden.py (the derived reference) again and compare member for member
    with what is committed.  config.json's \"raw\" is a list(set) in the reference (uq.py:899): order-insensitive."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLD, 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    names = []
    for name, fq, flags, stable in mg.cases():
        if name == 'cfg1_10k_100bp' and os.environ.get('UQ_SKIP_SLOW_GOLDEN'): continue
        mg.make_case(name, fq, flags, stable, outdir=str(tmp_path), refuses=mg.case_refuses(name))
        names.append(name)
        assert open(os.path.join(GOLD, name + '.fastq'), 'rb').read() == fq, name
        if mg.case_refuses(name):
            assert json.load(open(os.path.join(GOLD, name + '.json'))) == json.load(open(str(tmp_path / (name + '.json')))), name
            continue
        cfg_a, mem_a = O.read_tar(os.path.join(GOLD, name + '.uQ'))
        cfg_b, mem_b = O.read_tar(str(tmp_path / (name + '.uQ')))
        assert set(mem_a) == set(mem_b), name
        for k in mem_a: assert mem_a[k] == mem_b[k], (name, k)
        norm = lambda c: dict(c, raw=sorted(map(str, c['raw'])))
        assert norm(cfg_a) == norm(cfg_b), name
        ja, jb = json.load(open(os.path.join(GOLD, name + '.json'))), json.load(open(str(tmp_path / (name + '.json'))))
        assert ja == jb, name
        ra, rb = os.path.join(GOLD, name + '.refdecode.fastq'), str(tmp_path / (name + '.refdecode.fastq'))
        assert os.path.exists(ra) == os.path.exists(rb), name
        if os.path.exists(ra) and ja['reference_decode']['qname_lines'] == 'equal to the input':
            assert open(ra, 'rb').read() == open(rb, 'rb').read(), name
        elif os.path.exists(ra):      # Q6: QNAME column order follows os.listdir of a scratch directory -- compare SEQ / '+' / QUAL lines
            la, lb = open(ra, 'rb').read().split(b'\n'), open(rb, 'rb').read().split(b'\n')
            assert len(la) == len(lb) and all(la[k::4] == lb[k::4] for k in (1, 2, 3)), name
    assert sorted(names) == GOLDEN or os.environ.get('UQ_SKIP_SLOW_GOLDEN')


@pytest.mark.parametrize('name', [n for n in WRITTEN if n not in Q9])
def test_oracle_decode_roundtrip(name):
    fq = open(os.path.join(GOLD, name + '.fastq'), 'rb').read()
    ref_cfg, ref_members = O.read_tar(os.path.join(GOLD, name + '.uQ'))
    text = O.decode(ref_cfg, ref_members).encode('latin-1')
    rec = lambda b: sorted(b'\n'.join(x) for x in zip(*[iter(b.split(b'\n')[:-1])] * 4))
    if ref_cfg['sort'] == [None]: assert text == fq
    else: assert rec(text) == rec(fq)


def test_readme_geometry():
    """README.md:158-167, 287, 315-316: 5 symbols -> 3 bit, 36 bp -> 14 B; 26 quals -> 5 bit -> 23 B."""
    assert O.bits_for(5, False) == 3 and -(-3 * 36 // 8) == 14
    assert O.bits_for(26, False) == 5 and -(-5 * 36 // 8) == 23
    assert [O.bits_for(n, False) for n in (1, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 128, 129, 256)] == [2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8]
    assert [O.bits_for(n, True) for n in (4, 5, 16, 17, 41, 200)] == [2, 4, 4, 8, 8, 8]


def test_micro_vectors_a7():
    """SURVEY.md A.7, hand-checked."""
    def rows(reads, quals, **kw):
        fq = ''.join('@r:%d:%d\n%s\n+\n%s\n' % (i % 2, i, s, q) for i, (s, q) in enumerate(zip(reads, quals))).encode()
        return O.encode(fq, raw=['DNA', 'QUAL', 'QNAME'], **kw)
    # bases 'ACT' (G removed by the N-trick), b=2, read ACGTA -> codes 0,1,0,2,0 -> 0b0001001000 -> [0x00, 0x48]
    cfg, m, t = rows(['ACGTA', 'ACATA', 'TCATC'], ['IHGHH', 'HIIIH', 'HHIHI'])
    assert cfg['bases'] == 'ACT' and cfg['N_qual'] == {'G': 0} and t['DNA'][0].tolist() == [0x00, 0x48]
    # ACGT, b=2, L=6
    reads = ['ACGTAC', 'TTTTTT', 'ACGTAC', 'GATCCA', 'AAAAAA', 'GATCCA', 'ACGTAC', 'CCCCCC']
    cfg, m, t = rows(reads, ['IIHH!#', 'IIIIII', '#####I', 'IHIHIH', '!!!!!!', 'IIIIII', 'HHHHHH', 'IIIIII'])
    assert t['DNA'].tolist() == [[1, 177], [15, 255], [1, 177], [8, 212], [0, 0], [8, 212], [1, 177], [5, 85]]
    fq = ''.join('@r:%d:%d\n%s\n+\n%s\n' % (i % 2, i, s, 'IIHH!#' if i == 0 else 'IIIIII') for i, s in enumerate(reads)).encode()
    cfg, m, t = O.encode(fq, sort='DNA', pattern=['1.2', '0.1'])
    assert np.load(io.BytesIO(m['DNA.key'])).tolist() == [0, 1, 1, 1, 2, 3, 3, 4]
    assert np.load(io.BytesIO(m['DNA.key'])).dtype == np.uint8
    d = np.load(io.BytesIO(m['DNA']))
    assert d.shape == (2, 5) and d.tolist() == [[0, 177, 85, 212, 255], [0, 1, 5, 8, 15]] and d.flags.f_contiguous
    # variable length: ACGTACG -> [70,198]; TTN (N -> code 0) -> [_,124]; A -> [_,4] (the byte the reference never writes is 0, Q8)
    cfg, m, t = rows(['ACGTACG', 'TTN', 'A'], ['HIHIIHI', 'HI!', 'I'])
    assert cfg['variable_read_lengths'] and cfg['N_qual'] == {'N': 0} and cfg['qualities'] == '!HI'
    assert t['DNA'].tolist() == [[70, 198], [0, 124], [0, 4]]          # dna_max 7 + sentinel = 16 bits -> 2-byte rows
    assert t['QUAL'][1].tolist()[-1] == 88


def test_pattern_streams_a4():
    T = np.arange(12, dtype=np.uint8).reshape(4, 3)
    pay = lambda p: np.frombuffer(O.write_pattern(T, p), dtype=np.uint8)[-12:].tolist()
    assert pay('1.1') == [2, 5, 8, 11, 1, 4, 7, 10, 0, 3, 6, 9]
    assert pay('1.2') == [2, 1, 0, 5, 4, 3, 8, 7, 6, 11, 10, 9]
    assert pay('2.2') == [11, 8, 5, 2, 10, 7, 4, 1, 9, 6, 3, 0]
    assert pay('3.2') == [9, 10, 11, 6, 7, 8, 3, 4, 5, 0, 1, 2]
    for p in O.PATTERNS:
        assert np.array_equal(O.unpattern(O.write_pattern(T, p), p), T)


def test_encoder_loop_equals_closed_form():
    """The faithful flush loop (uq.py:147-175) == the closed form of SURVEY.md A.1/A.2, incl. Q7 lengths."""
    import random
    rnd = random.Random(5)
    for variable in (False, True):
        for bits in ((2, 6), (3, 5), (4, 8), (7, 2)):
            L = 9
            reads = [''.join(rnd.choice('ACGT') for _ in range(rnd.randint(1, L) if variable else L)) for _ in range(40)]
            if variable: reads[0] = 'A' * L; reads[1] = 'C' * 4; reads[2] = 'G' * 8
            lines = []
            for i, r in enumerate(reads): lines += ['@x\n', r + '\n', '+\n', ''.join(rnd.choice('!#%I') for _ in r) + '\n']
            lv = L + (1 if variable else 0)
            cd, cq = -(-bits[0] * lv // 8), -(-bits[1] * lv // 8)
            a = O.encoder(lines, 'ACGT', '!#%I', {}, cd, cq, bits[0], bits[1], variable)
            b = O.encoder_bigint(lines, 'ACGT', '!#%I', {}, cd, cq, bits[0], bits[1], variable)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
