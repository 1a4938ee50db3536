"""CPU: host logic of the product (no GPU, no compute calls): the C ABI loads and exports every symbol
include/uqhip.h declares, decisions (analysis.py) and QNAME passes (qname.py) agree with the oracle,
the .npy headers equal numpy's, CLI validation mirrors uq.py:52-71, the synthetic generator shards."""
import ctypes
import os
import re

import numpy as np
import pytest

import uq_oracle as O
from uq_amd import analysis, qname, synth, uq

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_header_symbols():
    from uq_amd import _lib, build
    build.build_lib()
    lib = _lib.load()
    assert _lib.MISSING == []
    header = open(os.path.join(REPO, 'include', 'uqhip.h')).read()
    declared = set(re.findall(r'^(?:int|const char\*)\s+(uq_[a-z0-9_]+)\s*\(', header, re.M))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
    assert declared - {'uq_last_error'} == set(_lib.SIGNATURES)
    assert lib.uq_abi_version() == _lib.ABI_VERSION
    assert ctypes.sizeof(_lib.Stats) == 65536 * 8 + 32
    assert lib.uq_key_itemsize(0) == 1 and lib.uq_key_itemsize(255) == 1 and lib.uq_key_itemsize(256) == 2
    assert lib.uq_key_itemsize(65536) == 4 and lib.uq_key_itemsize(1 << 32) == 8


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available(): pytest.skip('GPU present')
    from uq_amd._lib import UqHipError
    from uq_amd.device import Context
    with pytest.raises(UqHipError):
        Context(0)


@pytest.mark.parametrize('n,length,kw,dk', [(400, 100, {}, {}), (600, (36, 151), dict(n_rate=1), {}), (300, 50, dict(n_rate=3, n_qual_exclusive=False), {}),
                                             (300, (20, 60), dict(n_rate=2), dict(notricks=True)), (300, (20, 60), dict(n_rate=2), dict(notricks=True, pad=True))])
def test_decisions_match_oracle(n, length, kw, dk):
    lines = O.read_lines(synth.fastq(9, n, length, **kw))
    p1 = O.pass1(lines)
    od = O.decide(p1['static_qualities'], p1['dna_min'], p1['dna_max'], **dk)
    counts = np.zeros((256, 256), dtype=np.uint64)
    for b, qs in p1['static_qualities'].items():
        for q, c in qs.items(): counts[ord(b), ord(q)] = c
    d = analysis.decide_from_counts(counts, p1['dna_min'], p1['dna_max'], **dk)
    for k in ('bases', 'qualities', 'N_qual', 'bits_per_base', 'bits_per_quality', 'variable_read_lengths', 'dna_max',
              'dna_bytes_per_row', 'quality_bytes_per_row'):
        assert d[k] == od[k], k
    assert d['base_distribution'] == dict(od['base_graph']) and d['qual_distribution'] == dict(od['qual_graph'])


def test_n_trick_candidate_order_follows_first_appearance():
    # two bases with a single (shared) quality each: the order decides which new code each gets (Q11)
    counts = np.zeros((256, 256), dtype=np.uint64)
    for b in 'ACGT': counts[ord(b), ord('I')] = 10; counts[ord(b), ord('#')] = 5
    counts[ord('N'), ord('#')] = 3; counts[ord('X'), ord('#')] = 2
    fs = np.full(256, np.iinfo(np.uint64).max, dtype=np.uint64); fs[ord('X')] = 1; fs[ord('N')] = 9
    d = analysis.decide_from_counts(counts, 5, 5, first_seen=fs)
    assert d['N_qual'] == {'X': 3, 'N': 4}
    d = analysis.decide_from_counts(counts, 5, 5, first_seen=lambda: None)
    assert d['N_qual'] == {'N': 3, 'X': 4}


def test_qname_passes_match_oracle():
    for fq in (synth.fastq(5, 3000, 50), synth.fastq(6, 25000, 8),
               b''.join(b'@m%d#x:%d:%s#1\nA\n+\nI\n' % (i % 7, i % 300, b'ab' if i % 3 else b'c') for i in range(500))):
        lines = O.read_lines(fq)
        p1 = O.pass1(lines)
        names = [l[:-1] for l in lines[0::4]]
        pre, suf, sep, cols, arrs = qname.analyse(names)
        assert (pre, suf, sep) == (p1['prefix'], p1['suffix'], p1['separators'])
        ocols = O.qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
        assert cols == ocols
        oarr = O.qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], ocols)
        assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(arrs, oarr))
        cfg = {'QNAME_columns': cols, 'QNAME_separators': sep, 'QNAME_prefix': pre, 'QNAME_suffix': suf}
        assert qname.decode_names(cfg, arrs) == names


def test_qname_native_host_path_matches_python():
    """uq_qname_analyse (C++ on the host, row f1) == uq_amd/qname.py == the oracle, statement for statement."""
    import oracle_c
    cases = [synth.fastq(5, 3000, 50), synth.fastq(6, 25000, 8), open(os.path.join(REPO, 'tests', 'golden', 'cfg1_10k_100bp.fastq'), 'rb').read(),
             b''.join(b'@m%d#x:%d:%s#1\nA\n+\nI\n' % (i % 7, i % 300, b'ab' if i % 3 else b'c') for i in range(500)),
             b''.join(b'@r %d/%d\nA\n+\nI\n' % (i * 7919 % 100000, i % 2 + 1) for i in range(30000)),
             b''.join(b'@x:%d: %03d:%d\nA\n+\nI\n' % (i % 3, i % 500, i * 1000003 % 5000000000) for i in range(2500))]
    for fq in cases:
        host = np.frombuffer(fq, dtype=np.uint8)
        ls = oracle_c.index_lines(host)
        n = (len(ls) - 1) // 4
        lines = O.read_lines(fq)
        p1 = O.pass1(lines)
        ocols = O.qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
        oarr = O.qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], ocols)
        got = qname.analyse_native(host, ls, n)
        assert got is not None
        assert got[:3] == (p1['prefix'], p1['suffix'], p1['separators'])
        assert got[3] == ocols
        assert all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(got[4], oarr))
    # regex-special separators are handed to the Python implementation, hopeless inputs are refused like the reference does
    host = np.frombuffer(b'@a.1.x\nA\n+\nI\n@a.2.y\nA\n+\nI\n@a.3.z\nA\n+\nI\n', dtype=np.uint8)
    assert qname.analyse_native(host, oracle_c.index_lines(host), 3) is None
    host = np.frombuffer(b'@r1\nA\n+\nI\n@r2\nA\n+\nI\n', dtype=np.uint8)
    with pytest.raises(qname.QnameError):
        qname.analyse_native(host, oracle_c.index_lines(host), 2)


def test_qname_without_separators_is_refused():
    with pytest.raises(qname.QnameError):
        qname.analyse(['@r1', '@r2', '@r3'])


def test_npy_headers_equal_numpy():
    for (R, C) in [(1, 1), (1, 7), (5, 1), (4, 3), (64, 38), (333, 227), (2, 2)]:
        T = np.random.RandomState(1).randint(0, 256, size=(R, C)).astype(np.uint8)
        for pat in O.PATTERNS:
            npy = O.write_pattern(T, pat)
            assert npy[:len(npy) - R * C] == uq.pattern_header(R, C, pat), (R, C, pat)
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        k = np.arange(10, dtype=dt)
        assert O.npy_bytes(k) == uq.npy_header(k.shape, False, k.dtype) + k.tobytes()


def test_cli_validation(tmp_path):
    f = tmp_path / 'x.fastq'; f.write_bytes(b'@a\nA\n+\nI\n')
    P = uq.build_parser()
    a = uq.validate_args(P.parse_args(['-i', str(f), '--sort', 'None', '--raw', 'DNA', 'none']))
    assert a.sort == (None,) and a.raw == {'DNA', None}
    assert uq.validate_args(P.parse_args(['-i', str(f), '--sort', 'dna'])).sort == 'dna'     # Q1: accepted, matched case-sensitively later
    for bad in (['--pattern', '0.1'], ['--pattern', '0.1', '9.9'], ['--sort', 'x'], ['--raw', 'x']):
        with pytest.raises(uq.UqError):
            uq.validate_args(P.parse_args(['-i', str(f)] + bad))
    with pytest.raises(uq.UqError):
        uq.validate_args(P.parse_args(['-i', str(tmp_path / 'missing')]))
    flags = {a.option_strings[-1] for a in P._actions if a.option_strings}
    assert {'--input', '--output', '--compressor', '--sort', '--raw', '--pattern', '--temp', '--test', '--notricks', '--pad', '--peek', '--decode'} <= flags


def test_synth_is_stateless_and_shardable():
    spec = synth.Spec(3, (36, 90), n_rate=2, dup='dna', dup_templates=5)
    whole = synth.fastq_array(spec, 300)
    parts = np.concatenate([synth.fastq_array(spec, 100, first=f) for f in (0, 100, 200)])
    assert np.array_equal(whole, parts)
    lines = whole.tobytes().split(b'\n')
    assert len(lines) == 1201 and all(l.startswith(b'@SIM001:42:FCX01:') for l in lines[0:1200:4])
    size, ln, x, y = synth.record_sizes(spec, np.arange(300, dtype=np.uint64))
    assert size.sum() == whole.size and ln.min() >= 36 and ln.max() <= 90


def _top_level_args(text, start):
    """arguments of the call whose '(' is at text[start]: split at top-level commas; returns (list, index after ')')"""
    depth, cur, out, i = 0, '', [], start
    while True:
        ch = text[i]
        if ch in '([{':
            depth += 1
            if depth > 1: cur += ch
        elif ch in ')]}':
            depth -= 1
            if depth == 0:
                if cur.strip(): out.append(cur.strip())
                return out, i + 1
            cur += ch
        elif ch == ',' and depth == 1:
            out.append(cur.strip()); cur = ''
        elif ch == '#':                           # comment to end of line
            while text[i + 1] != '\n': i += 1
        else:
            cur += ch
        i += 1


def test_integration_md_snippets_match_the_binding():
    """The ctypes stubs INTEGRATION.md shows a maintainer: every `call('uq_*', ...)` passes as many arguments as the
    entry point takes (uq_amd/_lib.py SIGNATURES == include/uqhip.h), and the structures it declares have the
    binding's field names in the binding's order.  Keeps the document from drifting away from the boundary."""
    import re
    from uq_amd import _lib
    doc = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', doc, flags=re.S)
    assert blocks
    seen = set()
    for code in blocks:
        tuples = {}
        for m in re.finditer(r'^(\w+) = \(', code, flags=re.M):
            tuples[m.group(1)], _ = _top_level_args(code, m.end() - 1)
        for m in re.finditer(r"call\('(uq_\w+)'\s*,", code):
            name = m.group(1)
            args, _ = _top_level_args(code, code.rfind('(', 0, m.start() + 5))
            args = args[1:]                                        # drop the name
            n = 0
            for a in args:
                n += len(tuples[a[1:]]) if a.startswith('*') else 1
            assert name in _lib.SIGNATURES, name
            assert n == len(_lib.SIGNATURES[name]), '%s: INTEGRATION.md passes %d arguments, the ABI takes %d' % (name, n, len(_lib.SIGNATURES[name]))
            seen.add(name)
        for m in re.finditer(r'class (\w+)\(C\.Structure\):\s*\n\s*_fields_ = \[', code):
            fields = re.findall(r"\('(\w+)'", code[m.end():code.index(')]', m.end()) + 1])
            ref = {'Stats': _lib.Stats, 'PackParams': _lib.PackParams, 'Layout': _lib.QnameLayoutResult}[m.group(1)]
            assert fields == [f[0] for f in ref._fields_], m.group(1)
    assert {'uq_pack', 'uq_qname_layout', 'uq_decode_fastq', 'uq_stats_accumulate'} <= seen
    # every entry point of the header is in the document's table
    hdr = open(os.path.join(REPO, 'include', 'uqhip.h')).read()
    for name in re.findall(r'\b(uq_\w+)\s*\(', hdr):
        if name in _lib.SIGNATURES or name == 'uq_last_error':
            assert '`%s`' % name in doc or '`%s' % name in doc or name in doc, name + ' missing from INTEGRATION.md'


def test_decisions_from_the_nonzero_counters_equal_the_dense_ones():
    """analysis.decide_from_pairs (what the product feeds from uq_stats_fetch_compact: the non-zero (base, quality) counters as a list, in
    any order) against the dense 256 x 256 form on random tables: N-trick candidates (one quality only), several of them ordered by first
    occurrence, counts beyond 2^32, --notricks / --pad; and HostStats built from the compact record gives the same dense table back."""
    import numpy as np
    from uq_amd import analysis, ops
    rng = np.random.default_rng(7)
    for trial in range(200):
        c = np.zeros((256, 256), np.int64)
        for b in rng.choice(256, int(rng.integers(1, 7)), replace=False):
            k = int(rng.integers(1, 3)) if rng.random() < 0.4 else int(rng.integers(2, 60))
            for q in rng.choice(np.arange(33, 120), k, replace=False): c[b][q] = int(rng.integers(1, 10 ** 12))
        fs = rng.permutation(256)
        flat = c.reshape(-1); keys = np.flatnonzero(flat); perm = rng.permutation(len(keys))
        for kw in (dict(), dict(notricks=True), dict(pad=True)):
            a = analysis.decide_from_counts(c, 10, 10 + trial % 2, first_seen=fs, **kw)
            b = analysis.decide_from_pairs(keys[perm], flat[keys][perm], 10, 10 + trial % 2, first_seen=fs, **kw)
            assert a == b
            assert a['base_distribution'] == {chr(x): int(c[x].sum()) for x in np.flatnonzero(c.sum(axis=1))}
        raw = np.zeros(1, dtype=ops.STATS_COMPACT_DTYPE)
        n = min(len(keys), ops.STATS_COMPACT_CAP)
        raw[0]['n'] = n; raw[0]['key'][:n] = keys[perm][:n]; raw[0]['count'][:n] = flat[keys][perm][:n]
        raw[0]['len_min'] = 10; raw[0]['len_max'] = 11; raw[0]['bad_plus'] = ops.UQ_NONE; raw[0]['bad_len'] = ops.UQ_NONE
        hs = ops.HostStats(raw[0])
        if n == len(keys):
            assert np.array_equal(hs.counts.astype(np.int64), c)
            assert analysis.decide_from_stats(hs, first_seen=fs) == analysis.decide_from_counts(c, 10, 11, first_seen=fs)


def test_isa_scan_reads_a_kernel_file():
    """tools/isa_scan.py (no GPU: hipcc cross-compiles): the per-kernel report of loads / stores / waits / spills comes out for a small source file."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, 'tools', 'isa_scan.py'), 'scan.hip', '--all'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert 'scan_tile_kernel' in r.stdout and 'vgpr' in r.stdout and 'lwl' in r.stdout
