"""Host-side decisions between pass 1 and pass 3: alphabets, the N-trick, bit
widths and row bytes, from the 256 x 256 (base, quality) count matrix the device produced.

Mirrors uq.py:448-457 (alphabets, ASCII-sorted), 477-494 (N-trick), 497-516 and 534-545 (widths and
bytes per row).  Negligible work (<= 65536 counters), so it stays on the host like the reference's.
"""
import numpy as np


def bits_for(n_symbols, pad):
    """The width ladder of uq.py:497-503 / 534-540."""
    if n_symbols <= 4: return 2
    if n_symbols <= 8 and not pad: return 3
    if n_symbols <= 16: return 4
    if n_symbols <= 32 and not pad: return 5
    if n_symbols <= 64 and not pad: return 6
    if n_symbols <= 128 and not pad: return 7
    return 8


def decide_from_counts(counts, len_min, len_max, notricks=False, pad=False, first_seen=None):
    """counts[base][qual] (256 x 256, any integer dtype) -> the encoder's parameters.

    `first_seen`: optional array[256] (or a callable returning one) giving the position of each base's
    first occurrence; it orders the N-trick candidates the way the reference's dict iteration does
    (uq.py:480 under pypy / py3: first appearance).  Only consulted when two or more bases qualify.
    """
    counts = np.asarray(counts).reshape(256, 256)
    base_tot = counts.sum(axis=1)
    dna_bases = np.flatnonzero(base_tot).tolist()                 # uq.py:456 sorted(keys)
    if not dna_bases:
        raise ValueError('no bases counted')
    live = counts[dna_bases]                                      # the few rows that hold anything
    qual_tot = live.sum(axis=0)
    quals = np.flatnonzero(qual_tot).tolist()                     # uq.py:457
    all_bases = list(dna_bases)
    N_qual = {}
    total_quals = len(quals)
    if not notricks:                                              # uq.py:479-494
        nq = (live != 0).sum(axis=1)
        cand = [b for b, k in zip(dna_bases, nq.tolist()) if k == 1]
        if len(cand) > 1:
            fs = first_seen() if callable(first_seen) else first_seen
            if fs is not None:
                cand.sort(key=lambda b: int(fs[b]))
        for b in cand:
            if len(dna_bases) == 1: continue
            dna_bases.remove(b)
            q = int(np.nonzero(counts[b])[0][0])
            if counts[b][q] == qual_tot[q]:
                N_qual[chr(b)] = quals.index(q)
            else:
                total_quals += 1
                N_qual[chr(b)] = total_quals                      # replicated
    bits_per_base = bits_for(len(dna_bases), pad)
    bits_per_quality = bits_for(total_quals, pad)
    variable = int(len_min) != int(len_max)                       # uq.py:512-513
    lv = int(len_max) + (1 if variable else 0)
    return {
        'bases': ''.join(chr(b) for b in dna_bases), 'qualities': ''.join(chr(q) for q in quals),
        'N_qual': N_qual, 'total_quals': total_quals,
        'bits_per_base': bits_per_base, 'bits_per_quality': bits_per_quality,
        'variable_read_lengths': variable, 'dna_max': int(len_max), 'dna_min': int(len_min),
        'dna_bytes_per_row': -(-bits_per_base * lv // 8), 'quality_bytes_per_row': -(-bits_per_quality * lv // 8),
        'base_distribution': {chr(b): int(base_tot[b]) for b in all_bases},
        'qual_distribution': {chr(q): int(qual_tot[q]) for q in quals},
    }
