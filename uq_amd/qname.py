"""QNAME passes, sequential form, on the host (the device form is uq_amd/qname_device.py and
falls back to this module for QNAMEs outside the subset it reproduces exactly).

The reference infers a common prefix / suffix and constant-count separator characters while it scans
the file (uq.py:348-352, 394-444), types each delimited column as mapping / integers (uq.py:555-678)
and encodes it (uq.py:717-736).  These are string heuristics with order-dependent state; they run on
the host here, over the QNAME lines only (the record index tells where they are), and hand integer
columns to the device for the sort / unique / gather work.
"""
import bisect
import re

import numpy as np


class QnameError(Exception):
    pass


def qname_lines(host_bytes, line_start, nreads):
    """The QNAME lines (without '\\n') as latin-1 strings.  `line_start`: uint64 offsets (host)."""
    mv = memoryview(host_bytes)
    ls = line_start
    return [bytes(mv[int(ls[4 * r]):int(ls[4 * r + 1]) - 1]).decode('latin-1') for r in range(nreads)]


def infer_layout(names):
    """uq.py:348-352, 394-413, 428-444 -> (prefix, suffix, separators-as-ordered-string)."""
    line1 = names[0]
    if not line1.startswith('@'):
        raise QnameError('ERROR: This does not look like a FASTA/FASTQ file! (first line does not start with @)')
    prefix = line1
    suffix = line1
    separators = {}
    not_separators = set()
    for qname in names[1:]:
        if not qname.startswith(prefix):
            for idx, character in enumerate(prefix):
                if character != qname[idx]:
                    for sep in prefix[idx:]:
                        if sep not in not_separators:
                            separators[sep] = separators.get(sep, 0) + 1
                    prefix = prefix[:idx]
                    break
        if not qname.endswith(suffix):
            for idx, character in enumerate(reversed(suffix)):
                if character != qname[-1 - idx]:
                    suffix = '' if idx == 0 else suffix[-idx:]
                    break
        if separators:
            tail = qname[len(prefix):]
            for sep in list(separators):
                if tail.count(sep) != separators[sep]:
                    del separators[sep]
                    not_separators.add(sep)
    for sep in list(separators):
        c = suffix.count(sep)
        if c:
            separators[sep] -= c
            if separators[sep] == 0:
                del separators[sep]
    if not separators:
        raise QnameError('ERROR: the QNAMEs share no constant-count separator; the reference cannot encode such '
                         'files either (SURVEY.md Q13)')
    last = names[-1]

    def order_seps(q):
        found = re.findall('([' + ''.join(separators) + ']+)', q[len(prefix):-1 - len(suffix)])
        return ''.join(found)

    if order_seps(last) != order_seps(line1):
        raise QnameError("ERROR: Sorry, the separators used in this file's QNAME/headers are so unusual/improbable "
                         "that the reference gives up; so does this implementation")
    return prefix, suffix, order_seps(last)


_LADDER = [(255, 'uint8'), (65535, 'uint16'), (4294967295, 'uint32'), (18446744073709551615, 'uint64')]


def _int(s):
    if '_' in s:            # Python 2's int() knows no digit separators
        raise ValueError(s)
    return int(s)


def split_fields(names, prefix, suffix, separators):
    """uq.py:557-565: re.split('(.*)'.join(separators), name[len(prefix) : len - len(suffix)])."""
    regex = re.compile('(.*)'.join(separators))
    start = len(prefix)
    if suffix:
        end = -len(suffix)
        return [re.split(regex, n[start:end]) for n in names]
    return [re.split(regex, n[start:]) for n in names]


def type_columns(fields):
    """uq.py:571-676: mapping -> integers demotion at 10 000, 20 000, 40 000 ... reads and at the end;
    final dtype by cardinality / range; integer columns that fit stay integers."""
    columns = []
    cols = None
    target = 10000

    def check_format(entries_read):
        for c in columns:
            if c['format'] == 'mapping' and len(c['map']) > entries_read // 10:
                try:
                    v = [_int(x) for x in c['map']]
                    c['min'] = min(v); c['max'] = max(v); c['format'] = 'integers'
                    del c['map']
                except ValueError:
                    raise QnameError('Encoding QNAMEs as strings has not been implimented yet.')

    entries_read = -1
    for entries_read, f in enumerate(fields):
        if cols is None:
            cols = len(f)
            columns = [{'name': 'QNAME_%d' % (i + 1), 'format': 'mapping', 'map': set()} for i in range(cols)]
        elif len(f) != cols:
            raise QnameError('Encoding QNAMEs as strings has not been implimented yet. (the delimiters guessed for '
                             'the QNAMEs do not split every QNAME into the same number of fields)')
        for c, v in zip(columns, f):
            if c['format'] == 'mapping':
                c['map'].add(v)
            else:
                try:
                    iv = _int(v)
                except ValueError:
                    raise QnameError('Encoding QNAMEs as strings has not been implimented yet.')
                if iv < c['min']: c['min'] = iv
                elif iv > c['max']: c['max'] = iv
        if entries_read == target:
            check_format(entries_read)
            target *= 2
    check_format(entries_read)

    for c in columns:
        if c['format'] == 'mapping':
            map_len = len(c['map'])
            for lim, dt in _LADDER:
                if map_len <= lim:
                    map_len = lim; c['dtype'] = dt
                    break
            try:
                v = [_int(x) for x in c['map']]
                if max(v) - min(v) <= map_len:
                    c['format'] = 'integers'; c['max'] = max(v); c['min'] = min(v)
                    c['offset'] = bool(min(v) < 0 or max(v) > map_len)
                    del c['map']
                else:
                    c['map'] = sorted(c['map'])
            except Exception:
                c['map'] = sorted(c['map'])
        else:
            int_len = c['max'] - c['min']
            for lim, dt in _LADDER:
                if int_len <= lim:
                    int_len = lim; c['dtype'] = dt
                    break
            c['offset'] = bool(c['min'] < 0 or c['max'] > int_len)
    return columns


def encode_columns(fields, columns):
    """uq.py:717-736: one numpy array per column (bisect into the sorted map, or int() [- min])."""
    n = len(fields)
    out = []
    for ci, c in enumerate(columns):
        if c['format'] == 'mapping':
            m = c['map']
            vals = [bisect.bisect_left(m, f[ci]) for f in fields]
        elif c['offset']:
            mn = c['min']
            vals = [_int(f[ci]) - mn for f in fields]
        else:
            vals = [_int(f[ci]) for f in fields]
        out.append(np.asarray(vals, dtype=np.dtype(c['dtype'])) if n else np.zeros(0, dtype=c['dtype']))
    return out


def analyse_native(host_bytes, line_start, nreads):
    """The same passes through libuqhip.so's host-side C++ (uq_qname_analyse).  Returns the 5-tuple of
    analyse(), or None when the native code asks for the Python implementation (status 1)."""
    import ctypes as C
    import json
    from ._lib import call
    buf = np.ascontiguousarray(host_bytes)
    ls = np.ascontiguousarray(line_start, dtype=np.uint64)
    q = C.c_void_p(); status = C.c_int()
    call('uq_qname_analyse', buf.ctypes.data_as(C.c_void_p), ls.ctypes.data_as(C.c_void_p), int(nreads), C.byref(q), C.byref(status))
    if status.value == 1:
        return None
    if status.value != 0:
        from ._lib import load
        raise QnameError(load().uq_last_error().decode('latin-1'))
    try:
        js = C.c_char_p()
        call('uq_qname_json', q, C.byref(js))
        meta = json.loads(js.value.decode('ascii'))
        arrays = []
        for i, c in enumerate(meta['columns']):
            a = np.empty(nreads, dtype=np.dtype(c['dtype']))
            call('uq_qname_column', q, i, a.ctypes.data_as(C.c_void_p), a.nbytes)
            arrays.append(a)
    finally:
        call('uq_qname_free', q)
    return meta['prefix'], meta['suffix'], meta['separators'], meta['columns'], arrays


def analyse(names):
    """Passes 1 (QNAME part), 2 and 4 in one go -> (prefix, suffix, separators, columns, column arrays)."""
    prefix, suffix, separators = infer_layout(names)
    fields = split_fields(names, prefix, suffix, separators)
    columns = type_columns(fields)
    arrays = encode_columns(fields, columns)
    return prefix, suffix, separators, columns, arrays


def decode_names(config, column_arrays):
    """uq.py:1010-1024: prefix + fields joined by the separators + suffix, one string per read."""
    cols = config['QNAME_columns']
    seps = config['QNAME_separators']
    prefix, suffix = config['QNAME_prefix'], config['QNAME_suffix']
    parts = []
    for idx, c in enumerate(cols):
        a = column_arrays[idx]
        if c['format'] == 'mapping':
            m = c['map']
            strs = [m[int(v)] for v in a]
        else:
            off = c['min'] if c['offset'] else 0
            strs = [str(int(v) + off) for v in a]
        parts.append(strs)
    n = len(column_arrays[0]) if column_arrays else 0
    out = []
    for r in range(n):
        s = prefix
        for idx in range(len(cols)):
            s += parts[idx][r]
            if idx < len(seps): s += seps[idx]
        out.append(s + suffix)
    return out
