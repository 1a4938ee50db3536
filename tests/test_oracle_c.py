"""CPU: the C restatement (oracle/uq_oracle.c) against the Python restatement on seeded inputs."""
import numpy as np
import pytest

import oracle_c
import uq_oracle as O
from uq_amd import synth

CASES = [(300, 100, {}, {}), (500, (36, 151), dict(n_rate=1), {}), (400, (20, 77), dict(n_rate=2), dict(notricks=True)),
         (300, (1, 30), dict(n_rate=3), dict(notricks=True, pad=True)), (200, 50, dict(n_rate=3, n_qual_exclusive=False), {})]


@pytest.mark.parametrize('n,length,kw,dk', CASES)
def test_c_matches_python(n, length, kw, dk):
    host = synth.fastq_array(synth.Spec(77, length, **kw), n)
    lines = O.read_lines(host.tobytes())
    ls = oracle_c.index_lines(host)
    assert len(ls) == 4 * n + 1 and ls[-1] == host.size
    p1 = O.pass1(lines)
    st = oracle_c.stats(host, ls, 0, n)
    sq = O.histogram_to_static_qualities(st['counts'], st['first_seen'])
    assert {b: dict(v) for b, v in sq.items()} == {b: dict(v) for b, v in p1['static_qualities'].items()}
    assert list(sq) == list(p1['static_qualities'])                     # first-appearance order
    assert (st['len_min'], st['len_max']) == (p1['dna_min'], p1['dna_max'])
    d = O.decide(p1['static_qualities'], p1['dna_min'], p1['dna_max'], **dk)
    pd_, pq_ = O.encoder(lines, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'], d['quality_bytes_per_row'],
                         d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'])
    cd, cq, bad = oracle_c.pack(host, ls, 0, n, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                                d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
    assert bad is None and np.array_equal(cd, pd_) and np.array_equal(cq, pq_)
    if not (kw.get('n_qual_exclusive', True) is False):
        cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'],
                   bits_per_quality=d['bits_per_quality'], variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'])
        s, q, ln, bad = oracle_c.unpack(cd, cq, cfg)
        assert bad is None
        for r in range(n):
            assert s[r, :ln[r]].tobytes().decode() == lines[4 * r + 1][:-1]
            assert q[r, :ln[r]].tobytes().decode() == lines[4 * r + 3][:-1]


def test_c_patterns_match_numpy():
    from uq_amd.ops import PATTERN_IDS
    for (R, C) in [(1, 1), (1, 7), (5, 1), (4, 3), (64, 38), (33, 113)]:
        T = np.random.RandomState(R + C).randint(0, 256, size=(R, C)).astype(np.uint8)
        for pat in O.PATTERNS:
            payload = np.frombuffer(O.write_pattern(T, pat), dtype=np.uint8)[-R * C:]
            assert np.array_equal(payload, oracle_c.pattern(T, PATTERN_IDS[pat])), (R, C, pat)
