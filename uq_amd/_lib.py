"""ctypes binding of libuqhip.so (the C ABI declared in include/uqhip.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UQ_LIB_PATH') or os.path.join(HERE, 'libuqhip.so')       # UQ_LIB_PATH: A/B builds of the kernels (tuning)
UQ_NONE = (1 << 64) - 1
ABI_VERSION = 1


class UqHipError(RuntimeError):
    pass


class Stats(C.Structure):
    _fields_ = [('counts', C.c_uint64 * 65536), ('bad_plus', C.c_uint64), ('bad_len', C.c_uint64),
                ('len_min', C.c_uint32), ('len_max', C.c_uint32), ('max_record_bytes', C.c_uint32),
                ('reserved', C.c_uint32)]


class PackParams(C.Structure):
    _fields_ = [('dna_code', C.c_int16 * 256), ('qual_code', C.c_int16 * 256), ('n_qual', C.c_int32 * 256),
                ('bits_per_base', C.c_int32), ('bits_per_quality', C.c_int32), ('variable', C.c_int32),
                ('dna_bytes_per_row', C.c_int32), ('quality_bytes_per_row', C.c_int32),
                ('max_record_bytes', C.c_int32), ('dna_max', C.c_int32), ('avg_record_bytes', C.c_int32)]


class UnpackParams(C.Structure):
    _fields_ = [('base_char', C.c_uint8 * 256), ('qual_char', C.c_uint8 * 256), ('qual_n_base', C.c_uint8 * 256),
                ('bits_per_base', C.c_int32), ('bits_per_quality', C.c_int32), ('variable', C.c_int32),
                ('dna_bytes_per_row', C.c_int32), ('quality_bytes_per_row', C.c_int32), ('dna_max', C.c_int32)]


class EmitParams(C.Structure):
    _fields_ = [('prefix', C.c_uint8 * 256), ('suffix', C.c_uint8 * 256), ('separators', C.c_uint8 * 32),
                ('prefix_len', C.c_int32), ('suffix_len', C.c_int32), ('ncols', C.c_int32), ('dna_max', C.c_int32),
                ('itemsize', C.c_int32 * 32), ('add', C.c_int64 * 32)]


class QnameLayoutResult(C.Structure):
    _fields_ = [('min_lcp', C.c_uint32), ('min_lcs', C.c_uint32), ('flags', C.c_uint32), ('nch', C.c_uint32),
                ('entry', C.c_uint64 * 64), ('lastviol', C.c_uint64 * 64), ('ch', C.c_uint8 * 64)]


class QnameColsResult(C.Structure):
    _fields_ = [('first_nonint', C.c_uint64 * 32), ('vmin', C.c_int64 * 32), ('vmax', C.c_int64 * 32),
                ('any_long', C.c_uint32 * 32), ('flags', C.c_uint32), ('reserved', C.c_uint32)]


class QnameFused(C.Structure):
    _fields_ = [('ok', C.c_uint32), ('plen', C.c_uint32), ('slen', C.c_uint32), ('nsep', C.c_uint32), ('l1len', C.c_uint32),
                ('flags', C.c_uint32), ('sample_step', C.c_uint32), ('nth', C.c_uint32),
                ('line1', C.c_uint8 * 256), ('seps', C.c_uint8 * 32), ('inset', C.c_uint8 * 256),
                ('vmin', C.c_uint32 * 8), ('vmax', C.c_uint32 * 8), ('undetermined', C.c_uint32 * 8),
                ('nreads', C.c_uint64), ('thresholds', C.c_uint64 * 24), ('counts', (C.c_uint64 * 24) * 8)]


class SynthSpec(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('len_lo', C.c_int32), ('len_hi', C.c_int32), ('n_rate', C.c_int32),
                ('n_qual_exclusive', C.c_int32), ('dup', C.c_int32), ('dup_templates', C.c_int32),
                ('skip_len_mod4', C.c_int32), ('reserved', C.c_int32)]


_vp, _u64, _u32, _int, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
_P = C.POINTER

# name -> argtypes (every function returns int except uq_last_error)
SIGNATURES = {
    'uq_abi_version': [],
    'uq_device_count': [_P(_int)],
    'uq_ctx_create': [_int, _vp, _P(_vp)],
    'uq_ctx_destroy': [_vp],
    'uq_ctx_reserve': [_vp, _sz],
    'uq_ctx_sync': [_vp],
    'uq_dev_alloc': [_vp, _sz, _P(_vp)],
    'uq_dev_free': [_vp, _vp],
    'uq_h2d': [_vp, _vp, _vp, _sz],
    'uq_d2h': [_vp, _vp, _vp, _sz],
    'uq_memset': [_vp, _vp, _int, _sz],
    'uq_timer_start': [_vp],
    'uq_timer_stop': [_vp, _P(C.c_float)],
    'uq_count_lines': [_vp, _vp, _u64, _P(_u64)],
    'uq_count_lines_begin': [_vp, _vp, _u64],
    'uq_count_lines_chunk': [_vp, _vp, _u64, _u64, _u64],
    'uq_count_lines_end': [_vp, _vp, _u64, _P(_u64)],
    'uq_count_lines_end_async': [_vp, _vp, _u64],
    'uq_index_lines_async': [_vp, _vp, _u64, _u64, _vp],
    'uq_count_lines_wait': [_vp, _vp, _u64, _P(_u64), _P(_int)],
    'uq_index_lines': [_vp, _vp, _u64, _u64, _vp],
    'uq_stats_init': [_vp, _vp],
    'uq_stats_fetch': [_vp, _vp, _vp],
    'uq_stats_fetch_compact': [_vp, _vp, _vp],
    'uq_stats_accumulate': [_vp, _vp, _vp, _u64, _u64, _vp],
    'uq_stats_export': [_vp, _vp, _u32, _u32, _u64, _vp],
    'uq_stats_import': [_vp, _vp, _u32, _vp],
    'uq_first_occurrence': [_vp, _vp, _vp, _u64, _u64, _u64, _vp],
    'uq_pack': [_vp, _vp, _vp, _u64, _u64, _P(PackParams), _vp, _vp, _vp],
    'uq_pack_stats': [_vp, _vp, _vp, _u64, _u64, _P(PackParams), _vp, _vp, _vp, _vp, _P(_int)],
    'uq_pack_stats_async': [_vp, _vp, _vp, _u64, _P(PackParams), _vp, _vp, _vp, _vp, _P(_int)],
    'uq_pattern': [_vp, _vp, _u64, _u32, _int, _vp],
    'uq_unpattern': [_vp, _vp, _u64, _u32, _int, _vp],
    'uq_argsort_rows': [_vp, _vp, _u64, _u32, _vp],
    'uq_lower_bound_rows': [_vp, _vp, _u64, _u32, _vp, _u64, _vp],
    'uq_gather_rows': [_vp, _vp, _u64, _u32, _vp, _int, _u64, _vp],
    'uq_check_index_range': [_vp, _vp, _int, _u64, _u64, _P(_u64)],
    'uq_unique_rows': [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp, _P(_u64)],
    'uq_key_itemsize': [_u64],
    'uq_narrow': [_vp, _vp, _u64, _int, _vp],
    'uq_stack_columns': [_vp, _P(_vp), _P(_int), _int, _u64, _int, _vp],
    'uq_unstack_column': [_vp, _vp, _u64, _int, _int, _int, _int, _vp],
    'uq_unpack': [_vp, _vp, _vp, _u64, _P(UnpackParams), _vp, _vp, _vp, _vp],
    'uq_qname_analyse': [_vp, _vp, _u64, _P(_vp), _P(_int)],
    'uq_qname_json': [_vp, _P(C.c_char_p)],
    'uq_qname_column': [_vp, _int, _vp, _u64],
    'uq_qname_free': [_vp],
    'uq_qname_layout': [_vp, _vp, _vp, _u64, _u64, _vp, _u32, _P(QnameLayoutResult)],
    'uq_qname_tokenise': [_vp, _vp, _vp, _u64, _u32, _u32, _vp, _u32, _P(_vp), _P(_vp), _P(QnameColsResult)],
    'uq_prefix_distinct': [_vp, _vp, _int, _vp, _u64, _P(_u64), _int, _P(_u64)],
    'uq_int_prefix_distinct': [_vp, _vp, _u64, C.c_int64, _u64, _u64, _P(_u64), _int, _P(_u64)],
    'uq_encode_int': [_vp, _vp, _u64, C.c_int64, _int, _vp],
    'uq_unique_sorted_rows': [_vp, _vp, _u64, _u32, _vp, _vp, _P(_u64)],
    'uq_unique_rows_of_groups': [_vp, _vp, _u64, _u32, _vp, _vp, _u64, _vp],
    'uq_partition_order': [_vp, _vp, _u64, _u32, _vp, _vp, _P(_u64)],
    'uq_partition_rows': [_vp, _vp, _u32, _u32, _vp, _u64, _u64, _u64, _vp],
    'uq_owner_of_rows': [_vp, _vp, _u64, _P(C.c_int64), _u32, _vp],
    'uq_index_affine': [_vp, _vp, _int, _u64, C.c_int64, _vp, _int],
    'uq_invert_permutation': [_vp, _vp, _int, _u64, C.c_int64, _vp, _P(_u64)],
    'uq_scatter_rows': [_vp, _vp, _u64, _u32, _vp, _int, C.c_int64, _u64, _vp, _P(_u64)],
    'uq_qname_guess': [_vp, _vp, _vp, _u64, _vp],
    'uq_qname_guess_async': [_vp, _vp, _vp, _vp],
    'uq_pack_stats_qname': [_vp, _vp, _vp, _u64, _u64, _P(PackParams), _vp, _vp, _vp, _vp, _vp, _vp, _u64, _P(_int)],
    'uq_pack_stats_qname_async': [_vp, _vp, _vp, _u64, _P(PackParams), _vp, _vp, _vp, _vp, _vp, _vp, _u64, _P(_int)],
    'uq_qname_fused_finish': [_vp, _vp, _vp, _u64],
    'uq_qname_fused_fetch': [_vp, _vp, _P(QnameFused)],
    'uq_qname_fused_first_seen': [_vp, _vp, _u64, _u64, _u64, _P(_u32), _P(_u32), _int, _vp],
    'uq_encode_u32': [_vp, _vp, _u64, _u32, _int, _vp],
    'uq_encode_u32_columns': [_vp, _vp, _u64, _u64, _int, _P(_u32), _P(_int), _P(_vp)],
    'uq_emit_fastq': [_vp, _P(EmitParams), _P(_vp), _P(_vp), _P(_vp), _vp, _vp, _vp, _u64, _vp, _vp, _u64, _P(_u64)],
    'uq_debug_scribble_lds': [_vp, C.c_uint32],
    'uq_sort_config': [_vp, C.c_int64, _P(_int), _int],
    'uq_sort_counters': [_vp, _P(_u64), _P(_u64)],
    'uq_decode_fastq': [_vp, _P(EmitParams), _P(UnpackParams), _P(_vp), _P(_vp), _P(_vp), _vp, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _P(_u64), _P(_u64)],
    'uq_synth_size': [_vp, _P(SynthSpec), _u64, _u64, _P(_u64)],
    'uq_synth_fastq': [_vp, _P(SynthSpec), _u64, _u64, _vp, _u64],
}

_lib = None
MISSING = []


def load():
    """Load libuqhip.so (once).  Raises UqHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UqHipError('%s not found: build it with `python -m uq_amd.build` (hipcc, gfx950). '
                         'There is no CPU fallback.' % LIB_PATH)
    try:
        # one HIP runtime per process: torch ships its own libamdhip64; loading it first makes libuqhip.so bind to the
        # same one (loaded the other way round, the two runtimes disagree about the visible devices)
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.uq_last_error.restype = C.c_char_p
    lib.uq_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            MISSING.append(name)
            continue
        fn.argtypes = args
        fn.restype = C.c_int
    if lib.uq_abi_version() != ABI_VERSION:
        raise UqHipError('libuqhip.so ABI %d != binding ABI %d' % (lib.uq_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point; non-zero -> UqHipError(uq_last_error())."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise UqHipError('%s: %s' % (name, lib.uq_last_error().decode('utf-8', 'replace')))
    return rc
