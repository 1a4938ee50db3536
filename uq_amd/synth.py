"""synth-v1: stateless synthetic FASTQ.

Every character of read `i` is a pure function of (seed, i, slot), so any shard can be
generated on any GPU (`uq_synth_*` in csrc/synth.hip) or on the host (here, numpy) and the
two are byte-identical.  This is workload generation for tests and benchmarks, not part of
the encode path.

    value(i, s)  = splitmix64(seed * 2^40 + i * 1024 + s)
    slots 0..L-1       base      ('ACGT'[v & 3]; 'N' when n_rate > 0 and (v >> 8) % 100 < n_rate)
    slots 512..512+L-1 quality   ('!' + v % 41; with an exclusive N quality: N -> '!', others '"' + v % 40;
                                  with a shared N quality: N -> '#')
    slot 1020 / 1021   x, y      1000 + v % 29000
    slot 1022          length    lo + v % (hi - lo + 1)   (variable-length inputs)
    slot 1023          control   v % 10 == 0 -> the read copies template (v // 10) % T, i.e. takes its
                                 bases and/or qualities from virtual read 2^29 + template
    QNAME = @SIM001:42:FCX01:<1 + i % 4>:<1101 + i % 64>:<x>:<y>      line 3 = '+'
"""
import numpy as np

MASK = (1 << 64) - 1
TEMPLATE_BASE = 1 << 29
SLOT_QUAL = 512
SLOT_X, SLOT_Y, SLOT_LEN, SLOT_CTL = 1020, 1021, 1022, 1023
QNAME_PREFIX = b'@SIM001:42:FCX01:'
DUP_NONE, DUP_DNA, DUP_QUAL, DUP_BOTH = 0, 1, 2, 3
_DUP = {None: DUP_NONE, 'none': DUP_NONE, 'dna': DUP_DNA, 'qual': DUP_QUAL, 'both': DUP_BOTH}


def splitmix64(x):
    """splitmix64 output function on a uint64 array (wraps mod 2^64)."""
    with np.errstate(over='ignore'):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def value(seed, i, s):
    with np.errstate(over='ignore'):
        x = (np.uint64(seed) << np.uint64(40)) + np.asarray(i, dtype=np.uint64) * np.uint64(1024) + np.asarray(s, dtype=np.uint64)
    return splitmix64(x)


class Spec:
    """Generator parameters; `as_tuple()` is what the C-ABI generator takes."""

    def __init__(self, seed, length, n_rate=0, n_qual_exclusive=True, dup=None, dup_templates=None,
                 skip_len_mod4=False):
        self.seed = int(seed)
        if isinstance(length, (tuple, list)):
            self.len_lo, self.len_hi = int(length[0]), int(length[1])
        else:
            self.len_lo = self.len_hi = int(length)
        assert 1 <= self.len_lo <= self.len_hi <= 508
        self.n_rate = int(n_rate)
        self.n_qual_exclusive = bool(n_qual_exclusive)
        self.dup = _DUP[dup] if not isinstance(dup, int) else dup
        self.dup_templates = int(dup_templates) if dup_templates else 0
        if self.dup != DUP_NONE: assert self.dup_templates > 0
        self.skip_len_mod4 = bool(skip_len_mod4)


def lengths(spec, idx):
    if spec.len_lo == spec.len_hi:
        return np.full(len(idx), spec.len_lo, dtype=np.int64)
    v = value(spec.seed, idx, SLOT_LEN)
    ln = spec.len_lo + (v % np.uint64(spec.len_hi - spec.len_lo + 1)).astype(np.int64)
    if spec.skip_len_mod4:
        bad = (ln % 4) == 0
        up = np.where(ln + 1 <= spec.len_hi, ln + 1, ln - 1)
        ln = np.where(bad, up, ln)
    return ln


def _ndigits(v):
    return 1 + (v >= 10).astype(np.int64) + (v >= 100) + (v >= 1000) + (v >= 10000)


def record_sizes(spec, idx):
    """Bytes of each record and its parts: returns (size, len, x, y)."""
    idx = np.asarray(idx, dtype=np.uint64)
    ln = lengths(spec, idx)
    x = 1000 + (value(spec.seed, idx, SLOT_X) % np.uint64(29000)).astype(np.int64)
    y = 1000 + (value(spec.seed, idx, SLOT_Y) % np.uint64(29000)).astype(np.int64)
    # prefix + lane(1) ':' tile(4) ':' x ':' y '\n'  + seq '\n' + '+' '\n' + qual '\n'
    size = len(QNAME_PREFIX) + 1 + 1 + 4 + 1 + _ndigits(x) + 1 + _ndigits(y) + 1 + ln + 1 + 2 + ln + 1
    return size, ln, x, y


def fastq_array(spec, n, first=0):
    """uint8 array with records first .. first+n-1."""
    idx = np.arange(first, first + n, dtype=np.uint64)
    size, ln, x, y = record_sizes(spec, idx)
    off = np.concatenate([[0], np.cumsum(size)])
    out = np.empty(int(off[-1]), dtype=np.uint8)
    pos = off[:-1].copy()
    pre = np.frombuffer(QNAME_PREFIX, dtype=np.uint8)
    out[pos[:, None] + np.arange(len(pre))] = pre
    pos += len(pre)
    ii = idx.astype(np.int64)

    def put_char(p, c):
        out[p] = c

    def put_num(p, v):
        nd = _ndigits(v)
        for d in range(1, 6):
            m = nd == d
            if not m.any(): continue
            vv = v[m]; pp = p[m]
            for k in range(d):
                out[pp + k] = 48 + (vv // 10 ** (d - 1 - k)) % 10
        return p + nd

    pos = put_num(pos, 1 + ii % 4); put_char(pos, 58); pos += 1
    pos = put_num(pos, 1101 + ii % 64); put_char(pos, 58); pos += 1
    pos = put_num(pos, x); put_char(pos, 58); pos += 1
    pos = put_num(pos, y); put_char(pos, 10); pos += 1
    seq_off = pos
    qual_off = pos + ln + 3
    out[pos + ln] = 10; out[pos + ln + 1] = 43; out[pos + ln + 2] = 10; out[qual_off + ln] = 10

    src_d = idx.copy(); src_q = idx.copy()
    if spec.dup != DUP_NONE:
        ctl = value(spec.seed, idx, SLOT_CTL)
        isdup = (ctl % np.uint64(10)) == 0
        tmpl = np.uint64(TEMPLATE_BASE) + (ctl // np.uint64(10)) % np.uint64(spec.dup_templates)
        if spec.dup in (DUP_DNA, DUP_BOTH): src_d = np.where(isdup, tmpl, src_d)
        if spec.dup in (DUP_QUAL, DUP_BOTH): src_q = np.where(isdup, tmpl, src_q)

    # flat (read, slot) enumeration; chunked to bound memory
    CH = 1 << 16
    for a in range(0, n, CH):
        b = min(n, a + CH)
        l = ln[a:b]
        rec = np.repeat(np.arange(a, b), l)
        start = np.cumsum(l) - l
        slot = np.arange(int(l.sum())) - np.repeat(start, l)
        vb = value(spec.seed, src_d[rec], slot)
        vq = value(spec.seed, src_q[rec], SLOT_QUAL + slot)
        base = np.frombuffer(b'ACGT', dtype=np.uint8)[(vb & np.uint64(3)).astype(np.int64)]
        if spec.n_rate > 0:
            isn = ((vb >> np.uint64(8)) % np.uint64(100)) < np.uint64(spec.n_rate)
            base = np.where(isn, 78, base)
            if spec.n_qual_exclusive:
                q = np.where(isn, 33, 34 + (vq % np.uint64(40)).astype(np.int64))
            else:
                q = np.where(isn, 35, 33 + (vq % np.uint64(41)).astype(np.int64))
        else:
            q = 33 + (vq % np.uint64(41)).astype(np.int64)
        out[seq_off[rec] + slot] = base
        out[qual_off[rec] + slot] = q
    return out


def fastq(seed, n, length, first=0, **kw):
    """FASTQ bytes of `n` reads starting at read index `first`."""
    return fastq_array(Spec(seed, length, **kw), n, first).tobytes()
