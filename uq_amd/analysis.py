"""Host-side decisions between pass 1 and pass 3: alphabets, the N-trick, bit
widths and row bytes, from the 256 x 256 (base, quality) count matrix the device produced -- or from the list of its non-zero entries.

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
    flat = np.asarray(counts).reshape(65536)
    keys = np.flatnonzero(flat)
    return decide_from_pairs(keys, flat[keys], len_min, len_max, notricks=notricks, pad=pad, first_seen=first_seen)


def decide_from_stats(hs, notricks=False, pad=False, first_seen=None):
    """The same from a HostStats (ops.stats_fetch): its list of non-zero counters when it holds one, else its dense table."""
    if getattr(hs, 'nz_keys', None) is not None:
        return decide_from_pairs(hs.nz_keys, hs.nz_counts, hs.len_min, hs.len_max, notricks=notricks, pad=pad, first_seen=first_seen)
    return decide_from_counts(hs.counts, hs.len_min, hs.len_max, notricks=notricks, pad=pad, first_seen=first_seen)


def decide_from_pairs(keys, cnts, len_min, len_max, notricks=False, pad=False, first_seen=None):
    """The decisions from the NON-ZERO counters alone: keys = base * 256 + quality (any order), cnts their counts.  A file uses a
    few hundred of the 65 536 counters; everything below is arithmetic on that list."""
    keys = np.asarray(keys, dtype=np.int64); cnts = np.asarray(cnts, dtype=np.int64)
    live = cnts != 0
    if not live.all(): keys, cnts = keys[live], cnts[live]
    if keys.size == 0:
        raise ValueError('no bases counted')
    order = np.argsort(keys, kind='stable')
    keys, cnts = keys[order], cnts[order]
    b, q = keys >> 8, keys & 255
    # sums of at most 2^53 (a float64 holds them exactly: 9 * 10^15 symbols), far beyond a file
    base_tot = np.bincount(b, weights=cnts, minlength=256).astype(np.int64); qual_tot = np.bincount(q, weights=cnts, minlength=256).astype(np.int64)
    dna_bases = np.flatnonzero(base_tot).tolist()                 # uq.py:456 sorted(keys)
    quals = np.flatnonzero(qual_tot).tolist()                     # uq.py:457
    all_bases = list(dna_bases)
    N_qual = {}
    total_quals = len(quals)
    if not notricks:                                              # uq.py:479-494
        nq = np.bincount(b, minlength=256)                        # distinct qualities a base occurs with
        cand = [x for x in dna_bases if nq[x] == 1]
        if len(cand) > 1:
            fs = first_seen() if callable(first_seen) else first_seen
            if fs is not None:
                cand.sort(key=lambda x: int(fs[x]))
        for x in cand:
            if len(dna_bases) == 1: continue
            dna_bases.remove(x)
            at = int(np.searchsorted(b, x))                       # the base's one entry (keys are sorted: base-major)
            qq, c = int(q[at]), int(cnts[at])
            if c == int(qual_tot[qq]):
                N_qual[chr(x)] = quals.index(qq)
            else:
                total_quals += 1
                N_qual[chr(x)] = total_quals                      # replicated
    bits_per_base = bits_for(len(dna_bases), pad)
    bits_per_quality = bits_for(total_quals, pad)
    variable = int(len_min) != int(len_max)                       # uq.py:512-513
    lv = int(len_max) + (1 if variable else 0)
    return {
        'bases': ''.join(chr(x) for x in dna_bases), 'qualities': ''.join(chr(x) for x in quals),
        'N_qual': N_qual, 'total_quals': total_quals,
        'bits_per_base': bits_per_base, 'bits_per_quality': bits_per_quality,
        'variable_read_lengths': variable, 'dna_max': int(len_max), 'dna_min': int(len_min),
        'dna_bytes_per_row': -(-bits_per_base * lv // 8), 'quality_bytes_per_row': -(-bits_per_quality * lv // 8),
        'base_distribution': {chr(x): int(base_tot[x]) for x in all_bases},
        'qual_distribution': {chr(x): int(qual_tot[x]) for x in quals},
    }
