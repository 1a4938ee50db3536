"""QNAME passes with the per-read work on the device (device form).

The reference's loop over QNAME lines (uq.py:394-444, 555-678, 717-736) keeps order-dependent state,
but every decision it takes is a function of a few reductions over the reads (see csrc/qname_dev.hip).
This module runs those reductions / the tokeniser through the C ABI and finishes the decisions on the
host, on a handful of numbers per column.  It reproduces the reference only for inputs inside an
exactly-characterised subset (ASCII fields, plain decimal integers, separators that are not regex
metacharacters, ...); outside it `analyse_device` returns None and the caller uses the sequential
host implementation (`qname.analyse_native` / `qname.analyse`).  The column arrays stay in HBM.
"""
import numpy as np

from . import ops
from ._lib import UQ_NONE, UqHipError
from .qname import QnameError, _LADDER

# '[' + seps + ']+' and '(.*)'.join(seps) are regexes in the reference: leave their metacharacters to `re`
REGEX_SPECIAL = frozenset('.^$*+?{}[]\\|()-')
_STRINGS = 'Encoding QNAMEs as strings has not been implimented yet.'


def _ladder(x):
    for lim, dt in _LADDER:
        if x <= lim:
            return lim, dt
    return None, None


def _fetch_bytes(ctx, d_buf, lo, hi):
    return ctx.to_numpy(d_buf[lo:hi]).tobytes()


def infer_layout_device(ctx, d_buf, d_ls, nreads):
    """uq.py:348-352, 394-413, 428-444 -> (prefix, suffix, separators) as latin-1 strings, or None."""
    head = _fetch_bytes(ctx, d_buf, 0, min(257, d_buf.numel()))
    nl = head.find(b'\n')
    if nl < 0:
        return None
    line1 = head[:nl]
    if not line1.startswith(b'@'):
        raise QnameError('ERROR: This does not look like a FASTA/FASTQ file! (first line does not start with @)')
    if len(line1) > 255:
        return None
    try:
        res = ops.qname_layout(ctx, d_buf, d_ls, nreads, line1)
    except UqHipError:
        return None                      # more than 64 distinct characters in line 1
    if res.flags:
        return None
    plen, slen = res.min_lcp, res.min_lcs
    l1 = line1.decode('latin-1')
    prefix = l1[:plen]
    suffix = l1[len(l1) - slen:] if slen else ''
    seps = set()
    for k in range(res.nch):
        if res.entry[k] != UQ_NONE and res.lastviol[k] < res.entry[k]:
            c = chr(res.ch[k])
            if l1[plen:].count(c) - suffix.count(c) != 0:
                seps.add(c)
    if not seps:
        raise QnameError('ERROR: the QNAMEs share no constant-count separator; the reference cannot encode such '
                         'files either (SURVEY.md Q13)')
    if seps & REGEX_SPECIAL or plen + slen > len(l1):
        return None
    o = ctx.to_numpy(d_ls[4 * (nreads - 1):4 * (nreads - 1) + 2], np.uint64)
    last = _fetch_bytes(ctx, d_buf, int(o[0]), int(o[1]) - 1).decode('latin-1')

    def order_seps(q):
        return ''.join(ch for ch in q[plen:-1 - slen] if ch in seps)

    if order_seps(last) != order_seps(l1):
        raise QnameError("ERROR: Sorry, the separators used in this file's QNAME/headers are so unusual/improbable "
                         "that the reference gives up; so does this implementation")
    separators = order_seps(last)
    if not 1 <= len(separators) <= 31:
        return None
    return prefix, suffix, separators


def type_and_encode_device(ctx, d_buf, d_ls, nreads, prefix, suffix, separators):
    """uq.py:555-678 + 717-736 -> (columns, device column tensors) or None."""
    n = nreads
    vals, strs, res = ops.qname_tokenise(ctx, d_buf, d_ls, n, len(prefix), len(suffix), separators.encode('latin-1'))
    if res.flags:
        return None
    ncols = len(separators) + 1
    thresholds = []
    t = 10000
    while t <= n - 1:
        thresholds.append(t); t *= 2
    if not thresholds or thresholds[-1] != n - 1:
        thresholds.append(n - 1)
    columns, arrays = [], []
    for c in range(ncols):
        if res.any_long[c] & 1:
            return None
        col = {'name': 'QNAME_%d' % (c + 1), 'format': 'mapping'}
        first_nonint = res.first_nonint[c]
        all_int = first_nonint == UQ_NONE
        vmin, vmax = res.vmin[c], res.vmax[c]
        perm, key, skey, uniq, nu = ops.unique_rows(ctx, strs[c], n, 8)
        counts = ops.prefix_distinct(ctx, perm, skey, n, thresholds)
        for T, cnt in zip(thresholds, counts):
            if cnt > T // 10:                               # check_format(): mapping -> integers
                if first_nonint <= T or not all_int:
                    raise QnameError(_STRINGS)
                col['format'] = 'integers'
                break
        if col['format'] == 'mapping':
            lim, dt = _ladder(nu)
            col['dtype'] = dt
            if all_int and vmax - vmin <= lim:
                col['format'] = 'integers'; col['max'] = vmax; col['min'] = vmin
                col['offset'] = bool(vmin < 0 or vmax > lim)
            else:
                if res.any_long[c] & 2:
                    return None                             # sorted map of strings longer than the 8-byte key
                rows = ctx.to_numpy(uniq).reshape(nu, 8)
                col['map'] = [s.decode('latin-1') for s in np.ascontiguousarray(rows).view('S8').ravel().tolist()]
        else:
            col['min'] = vmin; col['max'] = vmax
            lim, dt = _ladder(vmax - vmin)
            col['dtype'] = dt
            col['offset'] = bool(vmin < 0 or vmax > lim)
        isz = np.dtype(col['dtype']).itemsize
        if col['format'] == 'mapping':
            arrays.append(ops.narrow(ctx, key, isz))
        else:
            arrays.append(ops.encode_int(ctx, vals[c], col['min'] if col['offset'] else 0, isz))
        columns.append(col)
        vals[c] = strs[c] = None                            # release this column's staging as we go
    return columns, arrays


def analyse_device(ctx, d_buf, d_ls, nreads):
    """(prefix, suffix, separators, columns, device column tensors), or None -> use the host path."""
    lay = infer_layout_device(ctx, d_buf, d_ls, nreads)
    if lay is None:
        return None
    prefix, suffix, separators = lay
    out = type_and_encode_device(ctx, d_buf, d_ls, nreads, prefix, suffix, separators)
    if out is None:
        return None
    return prefix, suffix, separators, out[0], out[1]
