"""ctypes binding of oracle/liboracle.so (the C restatement) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
LIB = os.path.join(ORACLE_DIR, 'liboracle.so')
NONE = (1 << 64) - 1
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, 'uq_oracle.c')):
            subprocess.check_call(['make', '-C', ORACLE_DIR, '-s'])
        _lib = C.CDLL(LIB)
        _lib.uqo_index_lines.restype = C.c_uint64
        _lib.uqo_pack.restype = C.c_uint64
        _lib.uqo_unpack.restype = C.c_uint64
        _lib.uqo_stats.restype = None
        _lib.uqo_pattern.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def index_lines(buf):
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    n = lib().uqo_index_lines(_p(buf), C.c_uint64(buf.size), None, C.c_uint64(0))
    ls = np.zeros(n + 1, dtype=np.uint64)
    lib().uqo_index_lines(_p(buf), C.c_uint64(buf.size), _p(ls), C.c_uint64(n))
    return ls


def stats(buf, ls, first, n):
    counts = np.zeros(65536, dtype=np.uint64)
    first_seen = np.full(256, NONE, dtype=np.uint64)
    lmin = C.c_uint32(0xFFFFFFFF); lmax = C.c_uint32(0); rmax = C.c_uint32(0)
    bp = C.c_uint64(NONE); bl = C.c_uint64(NONE)
    lib().uqo_stats(_p(buf), _p(ls), C.c_uint64(first), C.c_uint64(n), _p(counts), _p(first_seen),
                    C.byref(lmin), C.byref(lmax), C.byref(rmax), C.byref(bp), C.byref(bl))
    return dict(counts=counts.reshape(256, 256), first_seen=first_seen, len_min=lmin.value, len_max=lmax.value,
                max_record_bytes=rmax.value, bad_plus=None if bp.value == NONE else bp.value,
                bad_len=None if bl.value == NONE else bl.value)


def luts(bases, qualities, N_qual):
    d = np.full(256, -1, dtype=np.int16); q = np.full(256, -1, dtype=np.int16); nq = np.full(256, -1, dtype=np.int32)
    for i, ch in enumerate(bases): d[ord(ch)] = i
    for i, ch in enumerate(qualities): q[ord(ch)] = i
    for ch, code in N_qual.items(): nq[ord(ch)] = code
    return d, q, nq


def pack(buf, ls, first, n, bases, qualities, N_qual, bpb, bpq, variable, Cd, Cq):
    d, q, nq = luts(bases, qualities, N_qual)
    dna = np.zeros((n, Cd), dtype=np.uint8); qual = np.zeros((n, Cq), dtype=np.uint8)
    bad = lib().uqo_pack(_p(buf), _p(ls), C.c_uint64(first), C.c_uint64(n), _p(d), _p(q), _p(nq), int(bpb), int(bpq),
                         int(bool(variable)), C.c_uint32(Cd), C.c_uint32(Cq), _p(dna), _p(qual))
    return dna, qual, (None if bad == NONE else bad)


def unpack(dna, qual, config):
    n = dna.shape[0]
    bc = np.zeros(256, dtype=np.uint8); qc = np.zeros(256, dtype=np.uint8); qn = np.zeros(256, dtype=np.uint8)
    for i, ch in enumerate(config['bases']): bc[i] = ord(ch)
    for i, ch in enumerate(config['qualities']): qc[i] = ord(ch)
    for ch, code in config['N_qual'].items():
        if 0 <= int(code) < 256: qn[int(code)] = ord(ch)
    dmax = config['dna_max']
    seq = np.zeros((n, dmax), dtype=np.uint8); qt = np.zeros((n, dmax), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32)
    dna = np.ascontiguousarray(dna); qual = np.ascontiguousarray(qual)
    bad = lib().uqo_unpack(_p(dna), _p(qual), C.c_uint64(n), _p(bc), _p(qc), _p(qn), int(config['bits_per_base']),
                           int(config['bits_per_quality']), int(bool(config['variable_read_lengths'])),
                           C.c_uint32(dna.shape[1]), C.c_uint32(qual.shape[1]), C.c_uint32(dmax), _p(seq), _p(qt), _p(ln))
    return seq, qt, ln, (None if bad == NONE else bad)


def pattern(table, pattern_id):
    table = np.ascontiguousarray(table, dtype=np.uint8)
    out = np.zeros(table.size, dtype=np.uint8)
    lib().uqo_pattern(_p(table), C.c_uint64(table.shape[0]), C.c_uint32(table.shape[1]), int(pattern_id), _p(out))
    return out
