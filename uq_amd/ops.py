"""Thin typed wrappers over the C ABI: torch tensors in, torch tensors out.  One function per
entry point of include/uqhip.h; the reference site each one replaces is cited there."""
import ctypes as C

import numpy as np

from ._lib import call, Stats, PackParams, UnpackParams, SynthSpec, UQ_NONE, load

PATTERN_IDS = {'0.1': 0, '0.2': 1, '1.1': 2, '1.2': 3, '2.1': 4, '2.2': 5, '3.1': 6, '3.2': 7}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ------------------------------------------------------------------ record index
def count_lines(ctx, buf):
    n = C.c_uint64()
    call('uq_count_lines', ctx.h, _p(buf), buf.numel(), C.byref(n))
    return n.value


class ChunkedCensus:
    """uq_count_lines over a buffer that is still being filled: `chunk(lo, n)` as each piece lands, `end()` = the line count."""

    def __init__(self, ctx, buf):
        self.ctx, self.buf = ctx, buf
        call('uq_count_lines_begin', ctx.h, _p(buf), buf.numel())

    def chunk(self, lo, n):
        call('uq_count_lines_chunk', self.ctx.h, _p(self.buf), self.buf.numel(), int(lo), int(n))

    def end(self):
        out = C.c_uint64()
        call('uq_count_lines_end', self.ctx.h, _p(self.buf), self.buf.numel(), C.byref(out))
        return out.value

    # the queued form: end_async() -> index_lines_async / pack_stats_async (they take the line count on the device) -> wait()
    def end_async(self):
        call('uq_count_lines_end_async', self.ctx.h, _p(self.buf), self.buf.numel())

    def wait(self):
        """(line count, ok): ok False = the index made meanwhile is not usable (list or capacity overflow)."""
        out, ok = C.c_uint64(), C.c_int(0)
        call('uq_count_lines_wait', self.ctx.h, _p(self.buf), self.buf.numel(), C.byref(out), C.byref(ok))
        return out.value, bool(ok.value)


def index_lines(ctx, buf, nlines):
    """int64 tensor [nlines + 1] of line start offsets (bit pattern of uint64)."""
    t = ctx.torch
    ls = t.empty(nlines + 1, dtype=t.int64, device=ctx.device)
    call('uq_index_lines', ctx.h, _p(buf), buf.numel(), nlines, _p(ls))
    return ls


def index_lines_async(ctx, buf, capacity_lines):
    """The record index behind ChunkedCensus.end_async(): int64 tensor [capacity_lines + 1]; entries [0, lines] are written."""
    t = ctx.torch
    ls = t.empty(capacity_lines + 1, dtype=t.int64, device=ctx.device)
    call('uq_index_lines_async', ctx.h, _p(buf), buf.numel(), capacity_lines, _p(ls))
    return ls


# ------------------------------------------------------------------ pass-1 statistics
STATS_DTYPE = np.dtype([('counts', np.uint64, (65536,)), ('bad_plus', np.uint64), ('bad_len', np.uint64), ('len_min', np.uint32),
                        ('len_max', np.uint32), ('max_record_bytes', np.uint32), ('reserved', np.uint32)])    # struct uq_stats
assert STATS_DTYPE.itemsize == C.sizeof(Stats)


STATS_COMPACT_CAP = 2048
STATS_COMPACT_DTYPE = np.dtype([('n', np.uint32), ('len_min', np.uint32), ('len_max', np.uint32), ('max_record_bytes', np.uint32), ('reserved', np.uint32),
                                ('pad', np.uint32), ('bad_plus', np.uint64), ('bad_len', np.uint64), ('key', np.uint32, (STATS_COMPACT_CAP,)),
                                ('count', np.uint64, (STATS_COMPACT_CAP,))])                              # struct uq_stats_compact


class HostStats:
    """Host copy of uq_stats: counts[256][256] + ranges + first bad records.  From uq_stats_fetch_compact it holds the non-zero
    counters as a list (nz_keys = base * 256 + quality, nz_counts) and builds the dense table only when somebody asks for it."""

    def __init__(self, s):
        """`s`: a ctypes `Stats`, a numpy record of STATS_DTYPE (same layout) or one of STATS_COMPACT_DTYPE."""
        if isinstance(s, Stats):
            s = np.frombuffer(s, dtype=STATS_DTYPE, count=1)[0]
        if 'key' in s.dtype.names:
            n = int(s['n'])
            self.nz_keys = s['key'][:n].astype(np.int64); self.nz_counts = s['count'][:n].astype(np.int64)
            self._counts = None
        else:
            self.nz_keys = self.nz_counts = None
            self._counts = s['counts'].reshape(256, 256).copy()
        bad_plus, bad_len = int(s['bad_plus']), int(s['bad_len'])
        self.bad_plus = None if bad_plus == UQ_NONE else bad_plus
        self.bad_len = None if bad_len == UQ_NONE else bad_len
        self.len_min, self.len_max = int(s['len_min']), int(s['len_max'])
        self.max_record_bytes = int(s['max_record_bytes'])
        self.incomplete = bool(s['reserved'])         # set by uq_pack_stats only: counts not usable

    @property
    def counts(self):
        if self._counts is None:
            c = np.zeros(65536, dtype=np.uint64)
            c[self.nz_keys] = self.nz_counts.astype(np.uint64)
            self._counts = c.reshape(256, 256)
        return self._counts


def stats_new(ctx):
    t = ctx.torch
    d = t.empty(C.sizeof(Stats), dtype=t.uint8, device=ctx.device)
    call('uq_stats_init', ctx.h, _p(d))
    return d


def stats_accumulate(ctx, d_stats, buf, line_start, first_read, nreads):
    call('uq_stats_accumulate', ctx.h, _p(buf), _p(line_start), first_read, nreads, _p(d_stats))


def index_and_stats(ctx, buf, nlines):
    """Record index + pass-1 statistics of the whole buffer (uq_index_lines + uq_stats_accumulate).
    Returns (line_start tensor, device uq_stats)."""
    ls = index_lines(ctx, buf, nlines)
    st = stats_new(ctx)
    if nlines >= 4:
        stats_accumulate(ctx, st, buf, ls, 0, nlines // 4)
    return ls, st


def stats_fetch(ctx, d_stats):
    raw = np.empty(1, dtype=STATS_COMPACT_DTYPE)
    call('uq_stats_fetch_compact', ctx.h, _p(d_stats), C.c_void_p(raw.ctypes.data))
    if int(raw[0]['n']) <= STATS_COMPACT_CAP:
        return HostStats(raw[0])
    raw = np.empty(1, dtype=STATS_DTYPE)              # more non-zero counters than the list holds: the whole table
    call('uq_stats_fetch', ctx.h, _p(d_stats), C.c_void_p(raw.ctypes.data))
    return HostStats(raw[0])


def first_occurrence(ctx, buf, line_start, first_read, nreads, index_base=0):
    t = ctx.torch
    d = t.full((256,), -1, dtype=t.int64, device=ctx.device)
    call('uq_first_occurrence', ctx.h, _p(buf), _p(line_start), first_read, nreads, index_base, _p(d))
    return d.cpu().numpy().view(np.uint64)


# ------------------------------------------------------------------ pack
def make_pack_params(bases, qualities, N_qual, bits_per_base, bits_per_quality, variable,
                     dna_bytes_per_row, quality_bytes_per_row, dna_max, max_record_bytes, avg_record_bytes=0):
    p = PackParams()
    dna_code = np.full(256, -1, dtype=np.int16); qual_code = np.full(256, -1, dtype=np.int16); n_qual = np.full(256, -1, dtype=np.int32)
    dna_code[np.frombuffer(bases.encode('latin-1'), dtype=np.uint8)] = np.arange(len(bases), dtype=np.int16)
    qual_code[np.frombuffer(qualities.encode('latin-1'), dtype=np.uint8)] = np.arange(len(qualities), dtype=np.int16)
    for ch, code in N_qual.items(): n_qual[ord(ch)] = int(code)
    C.memmove(p.dna_code, dna_code.ctypes.data, 512); C.memmove(p.qual_code, qual_code.ctypes.data, 512)
    C.memmove(p.n_qual, n_qual.ctypes.data, 1024)
    p.bits_per_base = bits_per_base; p.bits_per_quality = bits_per_quality
    p.variable = 1 if variable else 0
    p.dna_bytes_per_row = dna_bytes_per_row; p.quality_bytes_per_row = quality_bytes_per_row
    p.max_record_bytes = max_record_bytes; p.dna_max = dna_max; p.avg_record_bytes = int(avg_record_bytes)
    return p


def pack(ctx, buf, line_start, first_read, nreads, params, dna=None, qual=None):
    """Returns (dna uint8[nreads * C_dna], qual uint8[nreads * C_qual], bad read index or None)."""
    t = ctx.torch
    if dna is None: dna = t.empty(nreads * params.dna_bytes_per_row, dtype=t.uint8, device=ctx.device)
    if qual is None: qual = t.empty(nreads * params.quality_bytes_per_row, dtype=t.uint8, device=ctx.device)
    bad = t.empty(1, dtype=t.int64, device=ctx.device)
    call('uq_pack', ctx.h, _p(buf), _p(line_start), first_read, nreads, C.byref(params), _p(dna), _p(qual), _p(bad))
    return dna, qual, bad


def pack_stats(ctx, buf, line_start, first_read, nreads, guess, fq=None):
    """One pass: pack with the GUESSED parameters and accumulate uq_stats of the same reads (uq_pack_stats).
    Returns (dna, qual, bad, d_stats) or None when there is no fused kernel for this geometry.
    `fq`: a FusedQname whose guess has been queued (qname_guess): the kernel also tokenises the QNAME lines (uq_pack_stats_qname)."""
    t = ctx.torch
    dna = t.empty(nreads * guess.dna_bytes_per_row, dtype=t.uint8, device=ctx.device)
    qual = t.empty(nreads * guess.quality_bytes_per_row, dtype=t.uint8, device=ctx.device)
    bad = t.empty(1, dtype=t.int64, device=ctx.device)
    st = stats_new(ctx)
    fused = C.c_int(0)
    if fq is None:
        call('uq_pack_stats', ctx.h, _p(buf), _p(line_start), first_read, nreads, C.byref(guess), _p(dna), _p(qual), _p(bad), _p(st), C.byref(fused))
    else:
        call('uq_pack_stats_qname', ctx.h, _p(buf), _p(line_start), first_read, nreads, C.byref(guess), _p(dna), _p(qual), _p(bad), _p(st),
             _p(fq.q), _p(fq.vals), fq.pitch, C.byref(fused))
    return (dna, qual, bad, st) if fused.value else None


def pack_stats_async(ctx, buf, line_start, capacity_reads, guess, st=None, fq=None):
    """pack_stats of every read of `buf` behind ChunkedCensus.end_async() + index_lines_async: tables of capacity_reads rows (the
    caller narrows them once wait() has told it the count; `st`: a stats_new() made earlier, so that its initialisation is not queued
    between the index and the pack kernel).  line_start may be None: no index is expanded, the kernel takes the line starts
    from the census's newline lists (include/uqhip.h).  Returns (dna, qual, bad, d_stats) or None (no fused kernel)."""
    t = ctx.torch
    dna = t.empty(capacity_reads * guess.dna_bytes_per_row, dtype=t.uint8, device=ctx.device)
    qual = t.empty(capacity_reads * guess.quality_bytes_per_row, dtype=t.uint8, device=ctx.device)
    bad = t.empty(1, dtype=t.int64, device=ctx.device)
    if st is None: st = stats_new(ctx)
    fused = C.c_int(0)
    if fq is None:
        call('uq_pack_stats_async', ctx.h, _p(buf), _p(line_start), capacity_reads, C.byref(guess), _p(dna), _p(qual), _p(bad), _p(st), C.byref(fused))
    else:
        call('uq_pack_stats_qname_async', ctx.h, _p(buf), _p(line_start), capacity_reads, C.byref(guess), _p(dna), _p(qual), _p(bad), _p(st),
             _p(fq.q), _p(fq.vals), fq.pitch, C.byref(fused))
    return (dna, qual, bad, st) if fused.value else None


# ------------------------------------------------------------------ the QNAME passes inside the pack kernel
QF_MAXC = 8


class FusedQname:
    """Device side of the fused QNAME pass (include/uqhip.h, uq_qname_fused): `q` = the structure in HBM, `vals` = QF_MAXC columns
    of `pitch` uint32 field values.  Order of calls: qname_guess[_async] -> pack_stats[_async](fq=...) -> qname_fused_finish ->
    qname_fused_fetch (the only one that waits for the device)."""

    def __init__(self, ctx, capacity_reads):
        from ._lib import QnameFused
        t = ctx.torch
        self.pitch = (int(capacity_reads) + 3) & ~3        # columns start 16-byte aligned
        self.q = t.empty(C.sizeof(QnameFused), dtype=t.uint8, device=ctx.device)
        self.vals = t.empty(QF_MAXC * self.pitch, dtype=t.int32, device=ctx.device)

    def column(self, c, n):
        return self.vals[c * self.pitch:c * self.pitch + n]


def qname_guess(ctx, buf, line_start, nreads, fq):
    call('uq_qname_guess', ctx.h, _p(buf), _p(line_start), int(nreads), _p(fq.q))


def qname_guess_async(ctx, buf, line_start, fq):
    """The layout guess behind ChunkedCensus.end_async() [+ index_lines_async] (the read count is taken on the device; line_start None:
    the sample is drawn through the census's newline lists)."""
    call('uq_qname_guess_async', ctx.h, _p(buf), _p(line_start), _p(fq.q))


def qname_fused_finish(ctx, fq):
    """Queue the distinct-value counts of the columns behind the pack kernel (no host wait)."""
    call('uq_qname_fused_finish', ctx.h, _p(fq.q), _p(fq.vals), fq.pitch)


def qname_fused_fetch(ctx, fq):
    from ._lib import QnameFused
    out = QnameFused()
    call('uq_qname_fused_fetch', ctx.h, _p(fq.q), C.byref(out))
    return out


def qname_fused_first_seen(ctx, fq, n, read_offset, vmins, ranges):
    """First occurrences of the small-range columns of a shard as file-wide read numbers (uq_qname_fused_first_seen): int64[QF_MAXC * 4096]
    on the device, INT64_MAX = absent; ranges[c] = 0 skips column c."""
    t = ctx.torch
    k = len(vmins)
    first = t.empty(QF_MAXC * 4096, dtype=t.int64, device=ctx.device)
    call('uq_qname_fused_first_seen', ctx.h, _p(fq.vals), fq.pitch, int(n), int(read_offset), (C.c_uint32 * k)(*[int(x) for x in vmins]),
         (C.c_uint32 * k)(*[int(x) for x in ranges]), k, _p(first))
    return first


def encode_u32_columns(ctx, fq, n, subs, itemsizes):
    """The first len(subs) columns of a fused pass narrowed to their dtypes in one launch (uq_encode_u32_columns)."""
    t = ctx.torch
    k = len(subs)
    outs = [t.empty(n, dtype=getattr(t, _NARROW_DT[isz]), device=ctx.device) for isz in itemsizes]
    call('uq_encode_u32_columns', ctx.h, _p(fq.vals), fq.pitch, int(n), k, (C.c_uint32 * k)(*[int(x) for x in subs]),
         (C.c_int * k)(*[int(x) for x in itemsizes]), (C.c_void_p * k)(*[o.data_ptr() for o in outs]))
    return outs


def encode_u32(ctx, val, n, sub, itemsize):
    t = ctx.torch
    out = t.empty(n, dtype=getattr(t, _NARROW_DT[itemsize]), device=ctx.device)
    call('uq_encode_u32', ctx.h, _p(val), int(n), int(sub), itemsize, _p(out))
    return out


HEAD_BYTES_SMALL = 4 << 20     # the slice of the file the pack-and-count encoder's guess is taken from (8192 reads of <= 512 bytes)
HEAD_READS_INDEXED = 8192


def head_guess(ctx, buf, notricks=False, pad=False, head_bytes=None, head_reads=None):
    """uq_pack_params guessed from the head of the file itself: the multi-pass statistics (uq.py:366-425) of its first
    HEAD_READS reads -> the decisions of uq.py:448-545 on that sample.  Returns (params, reads per byte estimate) or None when
    the head holds no complete record.  A guess is only ever used speculatively: the caller verifies it against the
    statistics of the WHOLE file (uq_pack_stats counts them in the same pass)."""
    from . import analysis
    with ctx.scope():                 # a SideContext: its tensors are allocated on ITS stream; `buf` (complete on the main one) is adopted
        head = ctx.adopt(buf)[:head_bytes or HEAD_BYTES_SMALL]
        nl = count_lines(ctx, head)
        n = min(nl // 4, head_reads or HEAD_READS_INDEXED)
        if n == 0:
            return None
        ls = index_lines(ctx, head, nl)
        st = stats_new(ctx)
        stats_accumulate(ctx, st, head, ls, 0, n)
        hs = stats_fetch(ctx, st)
    if hs.bad_plus is not None or hs.bad_len is not None:
        return None
    d = analysis.decide_from_stats(hs, notricks=notricks, pad=pad)
    if d['N_qual'] and max(d['N_qual'].values()) >= len(d['qualities']): return None       # Q9 new-code files: exact kernel only
    # bytes of the n reads, from the head's own byte and line counts (a read-back of ls[4 n] through torch would queue behind
    # whatever runs on torch's stream -- the caller's census of the whole file -- and hold the host until that is done)
    span = max(int(head.numel() * (4 * n) // max(nl, 1)), 4 * n)
    p = make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                         d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes, avg_record_bytes=span // n)
    return p, n / span


def head_guess_indexed(ctx, buf, line_start, nreads, notricks=False, pad=False):
    """head_guess for a buffer whose record index exists already: the statistics of its first HEAD_READS_INDEXED reads -> decisions ->
    uq_pack_params, or None (malformed head, Q9 new-code alphabets: no speculative kernel)."""
    from . import analysis
    # 8192 reads: the statistics kernel flushes its tables once per workgroup, and a sample of 150 tiles keeps that to 150
    # workgroups (65 536 reads: 1024 workgroups, 0.15 ms of contended flushes for 22 MB of input)
    n = min(int(nreads), HEAD_READS_INDEXED)
    if n == 0:
        return None
    st = stats_new(ctx)
    stats_accumulate(ctx, st, buf, line_start, 0, n)
    hs = stats_fetch(ctx, st)
    if hs.bad_plus is not None or hs.bad_len is not None:
        return None
    d = analysis.decide_from_stats(hs, notricks=notricks, pad=pad)
    if d['N_qual'] and max(d['N_qual'].values()) >= len(d['qualities']): return None
    return make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                            d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes,
                            avg_record_bytes=buf.numel() // max(int(nreads), 1))


def same_pack_params(a, b):
    """Do two uq_pack_params describe the same encoding (everything but the tile-sizing hint)?"""
    if (a.bits_per_base, a.bits_per_quality, a.variable, a.dna_bytes_per_row, a.quality_bytes_per_row, a.dna_max) != \
       (b.bits_per_base, b.bits_per_quality, b.variable, b.dna_bytes_per_row, b.quality_bytes_per_row, b.dna_max):
        return False
    return bytes(a.dna_code) == bytes(b.dna_code) and bytes(a.qual_code) == bytes(b.qual_code) and bytes(a.n_qual) == bytes(b.n_qual)


def scribble_lds(ctx, pattern=0xA5A5A5A5):
    """Test aid (uq_debug_scribble_lds): overwrite every CU's LDS so that reads of never-written LDS show."""
    call('uq_debug_scribble_lds', ctx.h, pattern & 0xFFFFFFFF)


def bad_index(bad_tensor):
    v = int(bad_tensor.cpu().numpy().view(np.uint64)[0])
    return None if v == UQ_NONE else v


# ------------------------------------------------------------------ patterns
def pattern(ctx, table, rows, cols, pattern_id, out=None):
    t = ctx.torch
    if isinstance(pattern_id, str): pattern_id = PATTERN_IDS[pattern_id]
    if out is None: out = t.empty(rows * cols, dtype=t.uint8, device=ctx.device)
    call('uq_pattern', ctx.h, _p(table), rows, cols, pattern_id, _p(out))
    return out


def unpattern(ctx, payload, rows, cols, pattern_id, out=None):
    t = ctx.torch
    if isinstance(pattern_id, str): pattern_id = PATTERN_IDS[pattern_id]
    if out is None: out = t.empty(rows * cols, dtype=t.uint8, device=ctx.device)
    call('uq_unpattern', ctx.h, _p(payload), rows, cols, pattern_id, _p(out))
    return out


# ------------------------------------------------------------------ sort / gather / unique
def sort_config(ctx, msd_min_rows=0, level_bits=None):
    """uq_sort_config: from how many rows a sort's round 0 runs as the MSD partition (0 = default, < 0 = never) and, optionally, its levels' digits."""
    bits = list(level_bits or [])
    call('uq_sort_config', ctx.h, int(msd_min_rows), (C.c_int * max(1, len(bits)))(*bits) if bits else None, len(bits))


def sort_counters(ctx):
    """(sorts whose round 0 ran as the MSD partition, sorts that took the LSD passes) since the context was created."""
    a, b = C.c_uint64(), C.c_uint64()
    call('uq_sort_counters', ctx.h, C.byref(a), C.byref(b))
    return a.value, b.value


def argsort_rows(ctx, table, rows, cols):
    t = ctx.torch
    perm = t.empty(rows, dtype=t.int32, device=ctx.device)
    call('uq_argsort_rows', ctx.h, _p(table), rows, cols, _p(perm))
    return perm


def lower_bound_rows(ctx, sorted_table, rows, cols, probes, nprobes):
    """int64 tensor [nprobes]: first row index of the sorted table that is >= each probe row."""
    t = ctx.torch
    pos = t.empty(nprobes, dtype=t.int64, device=ctx.device)
    call('uq_lower_bound_rows', ctx.h, _p(sorted_table), rows, cols, _p(probes), nprobes, _p(pos))
    return pos


def gather_rows(ctx, table, table_rows, cols, index, n_out=None, out=None):
    t = ctx.torch
    if n_out is None: n_out = index.numel()
    if out is None: out = t.empty(n_out * cols, dtype=t.uint8, device=ctx.device)
    call('uq_gather_rows', ctx.h, _p(table), table_rows, cols, _p(index), index.element_size(), n_out, _p(out))
    return out


def check_index_range(ctx, index, limit):
    """Lowest position whose value is >= limit, or None (uq_check_index_range): stored keys / mapping columns are
    checked before they index a table."""
    bad = C.c_uint64()
    call('uq_check_index_range', ctx.h, _p(index), index.element_size(), index.numel(), int(limit), C.byref(bad))
    return None if bad.value == UQ_NONE else int(bad.value)


def unique_rows(ctx, table, rows, cols, want_key=True, want_sorted_key=True, want_unique=True):
    """Returns (perm i32[rows], key or None, sorted_key or None, unique rows or None, nunique)."""
    t = ctx.torch
    perm = t.empty(rows, dtype=t.int32, device=ctx.device)
    key = t.empty(rows, dtype=t.int32, device=ctx.device) if want_key else None
    skey = t.empty(rows, dtype=t.int32, device=ctx.device) if want_sorted_key else None
    uniq = t.empty(rows * cols, dtype=t.uint8, device=ctx.device) if want_unique else None
    nu = C.c_uint64()
    call('uq_unique_rows', ctx.h, _p(table), rows, cols, _p(perm), _p(key), _p(skey), _p(uniq), C.byref(nu))
    if uniq is not None: uniq = uniq[:nu.value * cols]
    return perm, key, skey, uniq, nu.value


def unique_sorted_rows(ctx, sorted_table, rows, cols, want_unique=True):
    """unique of a table that is already in memcmp order (uq_unique_sorted_rows): (group id per row i32[rows], unique rows or None, nunique)."""
    t = ctx.torch
    group = t.empty(rows, dtype=t.int32, device=ctx.device)
    uniq = t.empty(rows * cols, dtype=t.uint8, device=ctx.device) if want_unique else None
    nu = C.c_uint64()
    call('uq_unique_sorted_rows', ctx.h, _p(sorted_table), rows, cols, _p(group), _p(uniq), C.byref(nu))
    if uniq is not None: uniq = uniq[:nu.value * cols]
    return group, uniq, nu.value


def unique_rows_of_groups(ctx, table, rows, cols, group, nunique, perm=None):
    """The distinct rows of a table from what its sort left (uq_unique_rows_of_groups): uint8[nunique * cols].  perm = the sort's order when
    the table has not been moved into it (only the distinct rows are gathered), None for a table in sorted order."""
    uniq = ctx.torch.empty(nunique * cols, dtype=ctx.torch.uint8, device=ctx.device)
    call('uq_unique_rows_of_groups', ctx.h, _p(table), rows, cols, _p(perm), _p(group), nunique, _p(uniq))
    return uniq


def partition_order(ctx, dest, n, ndest):
    """Stable partition of positions 0 .. n - 1 by dest[position] (uq_partition_order): (order i32[n], counts int64[ndest] ON THE DEVICE)."""
    t = ctx.torch
    order = t.empty(n, dtype=t.int32, device=ctx.device)
    counts = t.empty(ndest, dtype=t.int64, device=ctx.device)
    call('uq_partition_order', ctx.h, _p(dest), n, ndest, _p(order), _p(counts), None)
    return order, counts


def key_itemsize(max_key):
    return load().uq_key_itemsize(int(max_key))


_NARROW_DT = {1: 'uint8', 2: 'int16', 4: 'int32', 8: 'int64'}


def narrow(ctx, key, itemsize):
    t = ctx.torch
    out = t.empty(key.numel(), dtype=getattr(t, _NARROW_DT[itemsize]), device=ctx.device)
    call('uq_narrow', ctx.h, _p(key), key.numel(), itemsize, _p(out))
    return out


def stack_columns(ctx, cols, common_itemsize):
    t = ctx.torch
    n = cols[0].numel()
    ptrs = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
    sizes = (C.c_int * len(cols))(*[c.element_size() for c in cols])
    rows = t.empty(n * len(cols) * common_itemsize, dtype=t.uint8, device=ctx.device)
    call('uq_stack_columns', ctx.h, ptrs, sizes, len(cols), n, common_itemsize, _p(rows))
    return rows


def unstack_column(ctx, rows, n, ncols, common_itemsize, col, out_itemsize):
    t = ctx.torch
    out = t.empty(n, dtype=getattr(t, _NARROW_DT[out_itemsize]), device=ctx.device)
    call('uq_unstack_column', ctx.h, _p(rows), n, ncols, common_itemsize, col, out_itemsize, _p(out))
    return out


# ------------------------------------------------------------------ QNAME passes on the device
def qname_layout(ctx, buf, line_start, nreads, line1, read_index_base=0):
    """`line1`: bytes of the first QNAME line of the whole file; `read_index_base`: file-wide number of this
    shard's first read.  Returns the QnameLayoutResult structure."""
    from ._lib import QnameLayoutResult
    res = QnameLayoutResult()
    l1 = (C.c_uint8 * len(line1)).from_buffer_copy(line1)
    call('uq_qname_layout', ctx.h, _p(buf), _p(line_start), nreads, int(read_index_base), l1, len(line1), C.byref(res))
    return res


def qname_tokenise(ctx, buf, line_start, nreads, prefix_len, suffix_len, separators):
    """Returns (vals [ncols int64 tensors], strs [ncols int64 tensors holding 8 text bytes], QnameColsResult)."""
    from ._lib import QnameColsResult
    t = ctx.torch
    ncols = len(separators) + 1
    vals = [t.empty(nreads, dtype=t.int64, device=ctx.device) for _ in range(ncols)]
    strs = [t.empty(nreads, dtype=t.int64, device=ctx.device) for _ in range(ncols)]
    pv = (C.c_void_p * ncols)(*[v.data_ptr() for v in vals])
    ps = (C.c_void_p * ncols)(*[s.data_ptr() for s in strs])
    seps = (C.c_uint8 * len(separators)).from_buffer_copy(separators)
    res = QnameColsResult()
    call('uq_qname_tokenise', ctx.h, _p(buf), _p(line_start), nreads, prefix_len, suffix_len, seps, len(separators), pv, ps, C.byref(res))
    return vals, strs, res


def prefix_distinct(ctx, perm, sorted_key, n, thresholds):
    """`perm`: int32 (local argsort) or int64 (file-wide indices) tensor, one entry per sorted position."""
    th = (C.c_uint64 * len(thresholds))(*thresholds)
    out = (C.c_uint64 * len(thresholds))()
    call('uq_prefix_distinct', ctx.h, _p(perm), perm.element_size(), _p(sorted_key), n, th, len(thresholds), out)
    return list(out)


def int_prefix_distinct(ctx, val, n, vmin, value_range, thresholds, index_base=0):
    """Distinct values among reads [0, T] of the first n entries of an int64 column, per threshold T (uq_int_prefix_distinct)."""
    th = (C.c_uint64 * len(thresholds))(*thresholds)
    out = (C.c_uint64 * len(thresholds))()
    call('uq_int_prefix_distinct', ctx.h, _p(val), int(n), int(vmin), int(value_range), int(index_base), th, len(thresholds), out)
    return list(out)


def encode_int(ctx, val, sub, itemsize):
    t = ctx.torch
    out = t.empty(val.numel(), dtype=getattr(t, _NARROW_DT[itemsize]), device=ctx.device)
    call('uq_encode_int', ctx.h, _p(val), val.numel(), int(sub), itemsize, _p(out))
    return out


# ------------------------------------------------------------------ unpack
def make_unpack_params(config):
    p = UnpackParams()
    bases, quals = config['bases'], config['qualities']
    for i, ch in enumerate(bases): p.base_char[i] = ord(ch)
    for i, ch in enumerate(quals): p.qual_char[i] = ord(ch)
    for ch, code in config['N_qual'].items():
        if 0 <= int(code) < 256: p.qual_n_base[int(code)] = ord(ch)
    variable = bool(config['variable_read_lengths'])
    lv = config['dna_max'] + (1 if variable else 0)
    p.bits_per_base = config['bits_per_base']; p.bits_per_quality = config['bits_per_quality']
    p.variable = 1 if variable else 0
    p.dna_bytes_per_row = -(-p.bits_per_base * lv // 8)
    p.quality_bytes_per_row = -(-p.bits_per_quality * lv // 8)
    p.dna_max = config['dna_max']
    return p


def unpack(ctx, dna, qual, nreads, params):
    t = ctx.torch
    seq = t.empty(nreads * params.dna_max, dtype=t.uint8, device=ctx.device)
    qtxt = t.empty(nreads * params.dna_max, dtype=t.uint8, device=ctx.device)
    ln = t.empty(nreads, dtype=t.int32, device=ctx.device)
    bad = t.empty(1, dtype=t.int64, device=ctx.device)
    call('uq_unpack', ctx.h, _p(dna), _p(qual), nreads, C.byref(params), _p(seq), _p(qtxt), _p(ln), _p(bad))
    return seq, qtxt, ln, bad


# ------------------------------------------------------------------ FASTQ text emit (decode)
def _emit_params(ctx, config, column_tensors):
    """config.json's QNAME layout -> (uq_emit_params, column / string-table pointer arrays, tensors to keep alive)."""
    from ._lib import EmitParams
    cols = config['QNAME_columns']
    if len(cols) > 32: raise ValueError('more than 32 QNAME columns')
    p = EmitParams()
    pre = config['QNAME_prefix'].encode('latin-1'); suf = config['QNAME_suffix'].encode('latin-1')
    seps = config['QNAME_separators'].encode('latin-1')
    if len(pre) > 256 or len(suf) > 256: raise ValueError('QNAME prefix / suffix longer than 256 bytes')
    for i, b in enumerate(pre): p.prefix[i] = b
    for i, b in enumerate(suf): p.suffix[i] = b
    for i, b in enumerate(seps[:32]): p.separators[i] = b
    p.prefix_len = len(pre); p.suffix_len = len(suf); p.ncols = len(cols); p.dna_max = config['dna_max']
    keep = []
    d_cols = (C.c_void_p * 32)(); d_chars = (C.c_void_p * 32)(); d_offs = (C.c_void_p * 32)()
    for i, c in enumerate(cols):
        p.itemsize[i] = column_tensors[i].element_size()
        p.add[i] = int(c['min']) if (c['format'] != 'mapping' and c.get('offset')) else 0
        d_cols[i] = column_tensors[i].data_ptr()
        if c['format'] == 'mapping':
            strs = [s.encode('latin-1') for s in c['map']]
            offs = np.zeros(len(strs) + 1, dtype=np.uint32)
            offs[1:] = np.cumsum([len(s) for s in strs])
            chars = ctx.to_device(np.frombuffer(b''.join(strs) + b'\0', dtype=np.uint8))
            offt = ctx.to_device(offs)
            keep += [chars, offt]
            d_chars[i] = chars.data_ptr(); d_offs[i] = offt.data_ptr()
            p.add[i] = len(strs)                                 # mapping columns: the table's length (codes are clamped to it on the device)
    return p, d_cols, d_chars, d_offs, keep


def emit_fastq(ctx, config, column_tensors, seq, qual, ln, nreads):
    """Device-side FASTQ text: returns a uint8 tensor with the whole decoded file.
    `column_tensors`: one device tensor per QNAME column (values per read, the column's dtype)."""
    t = ctx.torch
    p, d_cols, d_chars, d_offs, keep = _emit_params(ctx, config, column_tensors)
    offsets = t.empty(nreads + 1, dtype=t.int64, device=ctx.device)
    total = C.c_uint64()
    call('uq_emit_fastq', ctx.h, C.byref(p), d_cols, d_chars, d_offs, _p(seq), _p(qual), _p(ln), nreads, _p(offsets), None, 0, C.byref(total))
    out = t.empty(total.value, dtype=t.uint8, device=ctx.device)
    call('uq_emit_fastq', ctx.h, C.byref(p), d_cols, d_chars, d_offs, _p(seq), _p(qual), _p(ln), nreads, _p(offsets), _p(out), total.value, C.byref(total))
    del keep
    return out


def decode_fastq(ctx, config, column_tensors, dna, qual, nreads):
    """Packed DNA / QUAL tables + QNAME columns -> the FASTQ text (uint8 device tensor), in one pass over the rows
    (uq_decode_fastq).  Returns (text, bad): bad = lowest row without a length sentinel, or None."""
    t = ctx.torch
    p, d_cols, d_chars, d_offs, keep = _emit_params(ctx, config, column_tensors)
    up = make_unpack_params(config)
    offsets = t.empty(nreads + 1, dtype=t.int64, device=ctx.device)
    ln = t.empty(nreads if up.variable else 0, dtype=t.int32, device=ctx.device)
    d_bad = t.empty(1, dtype=t.int64, device=ctx.device)
    total = C.c_uint64(); bad = C.c_uint64()
    args = (ctx.h, C.byref(p), C.byref(up), d_cols, d_chars, d_offs, _p(dna), _p(qual), nreads, _p(ln) if up.variable else None, _p(offsets), _p(d_bad))
    call('uq_decode_fastq', *args, None, 0, C.byref(total), C.byref(bad))
    if bad.value != 2 ** 64 - 1: return None, int(bad.value)
    out = t.empty(total.value, dtype=t.uint8, device=ctx.device)
    call('uq_decode_fastq', *args, _p(out), total.value, C.byref(total), C.byref(bad))
    del keep
    return out, None


# ------------------------------------------------------------------ synthetic input
def synth_spec(spec):
    """uq_amd.synth.Spec -> C struct."""
    s = SynthSpec()
    s.seed = spec.seed; s.len_lo = spec.len_lo; s.len_hi = spec.len_hi; s.n_rate = spec.n_rate
    s.n_qual_exclusive = 1 if spec.n_qual_exclusive else 0
    s.dup = spec.dup; s.dup_templates = spec.dup_templates; s.skip_len_mod4 = 1 if spec.skip_len_mod4 else 0
    return s


def synth_fastq(ctx, spec, first, n):
    t = ctx.torch
    cs = synth_spec(spec)
    nbytes = C.c_uint64()
    call('uq_synth_size', ctx.h, C.byref(cs), first, n, C.byref(nbytes))
    out = t.empty(nbytes.value, dtype=t.uint8, device=ctx.device)
    call('uq_synth_fastq', ctx.h, C.byref(cs), first, n, _p(out), nbytes.value)
    return out
