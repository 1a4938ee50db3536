"""uq_oracle -- TEST INFRASTRUCTURE ONLY.

CPU restatement (Python 3 + numpy) of the encode/decode hot path of the reference
`uq.py` (JohnLonginotto/uq).  It exists to CHECK the HIP path; nothing under
`uq_amd/` may import it.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use it.

Parity status: PINNED (derived run).  The reference is a Python-2 script that needs
`cffi`; it cannot be imported unmodified in this image.  `tests/golden/make_golden.py`
executes it through stdlib `lib2to3` + a memory-only cffi shim (no arithmetic in the
shim) and the members it wrote are committed under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against them byte for byte.
The reference itself ships no tests or golden vectors, so that
derived run plus the README's worked geometry (README.md:158-167, 315-316) and the
hand-checked micro-vectors of SURVEY.md A.7 are the pins.

Every function cites the reference lines it follows as `uq.py:<lines>`.

Deliberate, documented deviations:
  Q8  bytes the variable-length encoder never writes are defined as 0.
  Q7  sentinel carries into the next byte when bits*len % 8 == 0 (the reference
      raises OverflowError there, so there is no reference output to match).
  Q11 N-trick candidates are visited in order of first appearance in the file
      (pypy / py3 dict order, the runtime the README documents), not CPython-2 hash order.
  Q17 ties under --sort are broken by file order (stable); numpy's default introsort
      order is an implementation accident.
  Q18 unique()'s inverse is ravel()ed (era behaviour).
  Q6/Q19/Q25 decoder works on codes, orders QNAME members numerically, reads via BytesIO.
"""
import bisect
import collections
import io
import json
import re
import tarfile

import numpy

PATTERNS = ['0.1', '1.1', '2.1', '3.1', '0.2', '1.2', '2.2', '3.2']


class UqError(Exception):
    """The reference prints a message and exit()s (uq.py:48-50); the restatement raises."""


# --------------------------------------------------------------------------- input
def read_lines(data):
    """Split FASTQ bytes into lines that keep their '\\n' (what `next(f)` yields, uq.py:342).
    Enforces the `wc -l` % 4 rule of uq.py:85-87."""
    if isinstance(data, (bytes, bytearray, memoryview)):
        text = bytes(data).decode('latin-1')
    else:
        text = data
    n_newlines = text.count('\n')
    if n_newlines % 4 != 0:
        raise UqError('ERROR: The FASTQ file provided contains' + str(n_newlines) + 'rows, which is not divisible by 4!')
    lines = text.split('\n')
    # A file that does not end in '\n' has a trailing partial line that `wc -l` never counted;
    # the reference then loses it (Q21).  Keep only counted lines.
    lines = [l + '\n' for l in lines[:n_newlines]]
    return lines


# --------------------------------------------------------------------------- pass 1
def pass1(lines):
    """uq.py:338-444 -- histogram of (base, quality) pairs, DNA length range and the QNAME
    prefix / suffix / separator inference.  `lines` as returned by read_lines()."""
    if len(lines) < 4:
        raise UqError('ERROR: empty input')
    line1, line2, line3, line4 = lines[0][:-1], lines[1][:-1], lines[2], lines[3][:-1]  # uq.py:342
    if not line1.startswith('@'):
        raise UqError('ERROR: This does not look like a FASTA/FASTQ file! (first line does not start with @)')
    prefix = line1                                                                       # uq.py:349-352
    suffix = line1[:]
    separators = collections.OrderedDict()
    not_separators = set()
    dna_min = len(line2)                                                                 # uq.py:355-357
    dna_max = len(line2)
    if not line3.startswith('+'):
        raise UqError('ERROR: This does not look like a FASTA/FASTQ file! (third line does not start with +)')
    if len(line2) != len(line4):
        raise UqError('ERROR: This does not look like a FASTA/FASTQ file! (SEQ and QUAL lines are not the same length)')
    static_qualities = collections.OrderedDict()                                         # uq.py:369-375
    for base, qual in zip(line2, line4):
        static_qualities.setdefault(base, collections.OrderedDict())
        static_qualities[base][qual] = static_qualities[base].get(qual, 0) + 1

    qname = line1  # Q14: the reference NameErrors on a 1-record file; the restatement uses line1.
    total = len(lines) // 4
    for r in range(1, total):                                                            # uq.py:378-425
        qname = lines[4 * r][:-1]
        dna = lines[4 * r + 1][:-1]
        if lines[4 * r + 2][0] != '+':
            raise UqError('ERROR: For entry' + str(r) + 'the third line does not start with +')
        qualities = lines[4 * r + 3][:-1]
        if len(dna) != len(qualities):
            raise UqError('ERROR: Length of DNA does not match the length of the quality scores for entry ' + str(r))
        if not qname.startswith(prefix):                                                 # uq.py:395-401
            for idx, character in enumerate(prefix):
                if character != qname[idx]:
                    for sep in prefix[idx:]:
                        if sep not in not_separators:
                            separators[sep] = separators.get(sep, 0) + 1
                    prefix = prefix[:idx]
                    break
        if not qname.endswith(suffix):                                                   # uq.py:403-408
            for idx, character in enumerate(reversed(suffix)):
                if character != qname[-1 - idx]:
                    suffix = '' if idx == 0 else suffix[-idx:]
                    break
        for sep in list(separators):                                                     # uq.py:410-413
            if qname[len(prefix):].count(sep) != separators[sep]:
                del separators[sep]
                not_separators.add(sep)
        if dna_max < len(dna): dna_max = len(dna)                                        # uq.py:416-417
        if dna_min > len(dna): dna_min = len(dna)
        for base, qual in zip(dna, qualities):                                           # uq.py:420-425
            sq = static_qualities.get(base)
            if sq is None:
                sq = static_qualities[base] = collections.OrderedDict()
            sq[qual] = sq.get(qual, 0) + 1

    for sep in list(separators):                                                         # uq.py:428-431
        if suffix.count(sep) != 0:
            separators[sep] -= suffix.count(sep)
            if separators[sep] == 0:
                del separators[sep]

    def order_seps(q):                                                                   # uq.py:433-436
        unordered = ''.join(separators)
        found = re.findall('([' + unordered + ']+)', q[len(prefix):-1 - len(suffix)])
        return ''.join(found)

    if len(separators) == 0:
        # Q13: the reference compiles '([]+)' -> re.error before any output exists.
        raise UqError('ERROR: no constant-count QNAME separator survives (reference raises re.error, Q13)')
    if order_seps(qname) == order_seps(line1):                                           # uq.py:438-444
        separators = order_seps(qname)
    else:
        raise UqError("ERROR: Sorry, the separators used in this file's QNAME/headers are so unusual/improbable")
    return {
        'static_qualities': static_qualities, 'dna_min': dna_min, 'dna_max': dna_max,
        'prefix': prefix, 'suffix': suffix, 'separators': separators, 'reads': total,
    }


def histogram_to_static_qualities(counts, first_seen=None):
    """Build the `static_qualities` mapping of uq.py:369-375 from a 256x256 count matrix
    (counts[base][qual]).  Keys are visited in order of first appearance (`first_seen[base]`
    = position of the first occurrence) when given, else ASCII order."""
    counts = numpy.asarray(counts)
    bases = [b for b in range(256) if counts[b].any()]
    if first_seen is not None:
        bases.sort(key=lambda b: int(first_seen[b]))
    out = collections.OrderedDict()
    for b in bases:
        out[chr(b)] = collections.OrderedDict((chr(q), int(counts[b][q])) for q in range(256) if counts[b][q])
    return out


# --------------------------------------------------------------------------- decisions
def bits_for(n_symbols, pad):
    """uq.py:497-503 and 534-540 (the same ladder for DNA and QUAL)."""
    if n_symbols <= 4: return 2
    if n_symbols <= 8 and not pad: return 3
    if n_symbols <= 16: return 4
    if n_symbols <= 32 and not pad: return 5
    if n_symbols <= 64 and not pad: return 6
    if n_symbols <= 128 and not pad: return 7
    return 8


def decide(static_qualities, dna_min, dna_max, notricks=False, pad=False):
    """uq.py:448-457 (alphabets), 477-494 (N-trick), 497-516 / 534-545 (widths, row bytes)."""
    base_graph = collections.OrderedDict()
    qual_graph = collections.OrderedDict()
    for base, counts_per_qual in static_qualities.items():                               # uq.py:450-452
        base_graph[base] = sum(counts_per_qual.values())
        for qual, count in counts_per_qual.items():
            qual_graph[qual] = qual_graph.get(qual, 0) + count
    dna_bases = sorted(base_graph.keys())                                                # uq.py:456-457
    quals = sorted(qual_graph.keys())
    N_qual = {}
    total_quals = len(quals)
    if notricks is False:                                                                # uq.py:479-494
        for base, result in static_qualities.items():
            if len(dna_bases) == 1: continue
            if len(result) == 1:
                dna_bases.remove(base)
                for quality, count in result.items():
                    if count == qual_graph[quality]:
                        N_qual[base] = quals.index(quality)
                    else:
                        total_quals += 1
                        N_qual[base] = total_quals                                       # Q9: replicated
    bits_per_base = bits_for(len(dna_bases), pad)
    variable_read_lengths = dna_min != dna_max                                           # uq.py:512-513
    dna_bits = bits_per_base * (dna_max + variable_read_lengths)
    dna_columns_needed = -(-dna_bits // 8)
    bits_per_quality = bits_for(total_quals, pad)
    qual_bits = bits_per_quality * (dna_max + variable_read_lengths)
    qual_columns_needed = -(-qual_bits // 8)
    return {
        'base_graph': base_graph, 'qual_graph': qual_graph,
        'bases': ''.join(dna_bases), 'qualities': ''.join(quals), 'N_qual': N_qual,
        'total_quals': total_quals,
        'bits_per_base': bits_per_base, 'bits_per_quality': bits_per_quality,
        'variable_read_lengths': variable_read_lengths, 'dna_max': dna_max, 'dna_min': dna_min,
        'dna_bytes_per_row': dna_columns_needed, 'quality_bytes_per_row': qual_columns_needed,
    }


# --------------------------------------------------------------------------- pass 3 (pack)
def encoder(lines, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row,
            bits_per_base, bits_per_quality, variable_read_lengths, first=0, count=None):
    """uq.py:108-182 (`encoder_fixed`) and 188-254 (`encoder_variable`): the faithful per-base
    loop, including the `+=` accumulation, the `> 8` flush rule and the reversed read.
    Returns (dna uint8[N][C_dna], qual uint8[N][C_qual]).  This loop is what bench.py times
    as the CPU baseline ("port")."""
    total_reads = len(lines) // 4 if count is None else count
    dna_array = numpy.zeros((total_reads, dna_bytes_per_row), dtype=numpy.uint8)        # Q8: zero, not malloc garbage
    qual_array = numpy.zeros((total_reads, quality_bytes_per_row), dtype=numpy.uint8)
    N_base = 0
    sentinel = 1 if variable_read_lengths else 0
    for row in range(total_reads):
        dna = lines[4 * (first + row) + 1][-2::-1]                                       # uq.py:135-137
        quals = lines[4 * (first + row) + 3][-2::-1]
        temp_dna = 0; temp_qual = 0; dna_bits_done = 0; qual_bits_done = 0
        dna_byte_position = dna_bytes_per_row - 1
        qual_byte_position = quality_bytes_per_row - 1
        drow = dna_array[row]; qrow = qual_array[row]
        for base in range(len(dna)):
            try:                                                                         # uq.py:148-153
                temp_dna += bases.index(dna[base]) << dna_bits_done
                temp_qual += qualities.index(quals[base]) << qual_bits_done
            except ValueError:
                temp_dna += N_base << dna_bits_done
                temp_qual += N_qual[dna[base]] << qual_bits_done
            dna_bits_done += bits_per_base
            qual_bits_done += bits_per_quality
            while dna_bits_done > 8:                                                     # uq.py:157-161
                dna_bits_done -= 8
                drow[dna_byte_position] = temp_dna & 255
                temp_dna >>= 8
                dna_byte_position -= 1
            while qual_bits_done > 8:                                                    # uq.py:163-167
                qual_bits_done -= 8
                qrow[qual_byte_position] = temp_qual & 255
                temp_qual >>= 8
                qual_byte_position -= 1
        # uq.py:170-171 (fixed) / 242-243 (variable).  Values that do not fit the byte carry
        # upward (Q7 rule; the reference's cffi store raises there) and are dropped at byte 0.
        temp_dna += sentinel << dna_bits_done
        temp_qual += sentinel << qual_bits_done
        while temp_dna and dna_byte_position >= 0:
            drow[dna_byte_position] = temp_dna & 255; temp_dna >>= 8; dna_byte_position -= 1
        while temp_qual and qual_byte_position >= 0:
            qrow[qual_byte_position] = temp_qual & 255; temp_qual >>= 8; qual_byte_position -= 1
        # uq.py:173-174 / 245-246: the rest stays zero (Q8).
    return dna_array, qual_array


def encoder_bigint(lines, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row,
                   bits_per_base, bits_per_quality, variable_read_lengths):
    """The closed form of SURVEY.md A.1/A.2 (value = sum code_j << b(L-1-j) [+ 1 << bL], stored
    big-endian, right-aligned, truncated to C bytes).  Used to cross-check `encoder`."""
    total_reads = len(lines) // 4
    dna_lut = {c: i for i, c in enumerate(bases)}
    qual_lut = {c: i for i, c in enumerate(qualities)}
    dna_array = numpy.zeros((total_reads, dna_bytes_per_row), dtype=numpy.uint8)
    qual_array = numpy.zeros((total_reads, quality_bytes_per_row), dtype=numpy.uint8)
    for row in range(total_reads):
        dna = lines[4 * row + 1][:-1]
        quals = lines[4 * row + 3][:-1]
        vd = 0; vq = 0
        for b, q in zip(dna, quals):
            if b in dna_lut:
                cd = dna_lut[b]; cq = qual_lut[q]
            else:
                cd = 0; cq = N_qual[b]
            vd = (vd << bits_per_base) + cd
            vq = (vq << bits_per_quality) + cq
        if variable_read_lengths:
            vd += 1 << (bits_per_base * len(dna))
            vq += 1 << (bits_per_quality * len(dna))
        vd &= (1 << (8 * dna_bytes_per_row)) - 1
        vq &= (1 << (8 * quality_bytes_per_row)) - 1
        dna_array[row] = numpy.frombuffer(vd.to_bytes(dna_bytes_per_row, 'big'), dtype=numpy.uint8)
        qual_array[row] = numpy.frombuffer(vq.to_bytes(quality_bytes_per_row, 'big'), dtype=numpy.uint8)
    return dna_array, qual_array


# --------------------------------------------------------------------------- pass 2 + 4 (QNAME)
_DTYPE_LADDER = [(255, 'uint8'), (65535, 'uint16'), (4294967295, 'uint32'), (18446744073709551615, 'uint64')]


def _py2_int(s):
    """int() as Python 2 parses it (no '_' digit separators)."""
    if '_' in s:
        raise ValueError(s)
    return int(s)


def qname_fields(lines, prefix, suffix, separators):
    """uq.py:557-570 -- `re.split('(.*)'.join(separators), line[len(prefix):-1-len(suffix)])`."""
    start = len(prefix)
    end = -1 - len(suffix)
    regex = re.compile('(.*)'.join(separators))
    for r in range(len(lines) // 4):
        yield re.split(regex, lines[4 * r][start:end])


def qname_columns(lines, prefix, suffix, separators):
    """uq.py:571-676 -- column typing (mapping -> integers -> strings) and dtype choice."""
    cols = False
    target = 10000
    columns = []

    def check_format(columns, entries_read):                                             # uq.py:586-602
        for column in columns:
            if column['format'] == 'mapping':
                if len(column['map']) > entries_read // 10:
                    try:
                        _ = [_py2_int(x) for x in column['map']]
                        column['min'] = min(_); column['max'] = max(_)
                        column['format'] = 'integers'
                        del column['map']
                    except ValueError:
                        column['format'] = 'strings'
                        column['longest'] = max(map(len, column['map']))
                        del column['map']

    entries_read = -1
    for entries_read, qname in enumerate(qname_fields(lines, prefix, suffix, separators)):
        if len(qname) != cols:                                                           # uq.py:605-613
            if cols is False:
                cols = len(qname)
                for col in range(cols):
                    columns.append({'name': 'QNAME_' + str(col + 1), 'format': 'mapping', 'map': set()})
            else:
                raise UqError('Encoding QNAMEs as strings has not been implimented yet.')
        for idx, column in enumerate(qname):                                             # uq.py:614-633
            c = columns[idx]
            if c['format'] == 'mapping':
                c['map'].add(column)
            elif c['format'] == 'integers':
                try:
                    column = _py2_int(column)
                    if column < c['min']: c['min'] = column
                    elif column > c['max']: c['max'] = column
                except ValueError:
                    c['format'] = 'strings'
                    c['longest'] = len(str(c['max']))
                    del c['min']; del c['max']
            elif c['format'] == 'strings':
                if len(column) > c['longest']: c['longest'] = len(column)
        if entries_read == target:                                                       # uq.py:634-636
            check_format(columns, entries_read)
            target *= 2
    check_format(columns, entries_read)                                                  # uq.py:638

    for column in columns:                                                               # uq.py:641-673
        if column['format'] == 'mapping':
            map_len = len(column['map'])
            for lim, dt in _DTYPE_LADDER:
                if map_len <= lim:
                    map_len = lim; column['dtype'] = dt
                    break
            try:
                _ = [_py2_int(x) for x in column['map']]
                if max(_) - min(_) <= map_len:
                    column['format'] = 'integers'
                    column['max'] = max(_); column['min'] = min(_)
                    column['offset'] = bool(min(_) < 0 or max(_) > map_len)
                    del column['map']
                else:
                    column['map'] = sorted(column['map'])
            except Exception:
                column['map'] = sorted(column['map'])
        elif column['format'] == 'integers':
            int_len = column['max'] - column['min']
            for lim, dt in _DTYPE_LADDER:
                if int_len <= lim:
                    int_len = lim; column['dtype'] = dt
                    break
            column['offset'] = bool(column['min'] < 0 or column['max'] > int_len)
        elif column['format'] == 'strings':
            raise UqError('I havent implimented this yet')
    return columns


def qname_encode(lines, prefix, suffix, separators, columns):
    """uq.py:717-736 -- pass 4: per-field bisect / int() into one array per column."""
    total = len(lines) // 4
    out = [numpy.zeros(total, dtype=c['dtype']) for c in columns]
    for row, qname in enumerate(qname_fields(lines, prefix, suffix, separators)):
        for column, data in enumerate(qname):
            c = columns[column]
            if c['format'] == 'mapping':
                out[column][row] = bisect.bisect_left(c['map'], data)
            elif c['format'] == 'integers' and c['offset']:
                out[column][row] = _py2_int(data) - c['min']
            elif c['format'] == 'integers':
                out[column][row] = _py2_int(data)
    return out


# --------------------------------------------------------------------------- tables
def npy_bytes(array):
    """`numpy.save(f, array)` (uq.py:263-274) to bytes."""
    f = io.BytesIO()
    numpy.save(f, array)
    return f.getvalue()


def pattern_array(table, pattern):
    """uq.py:263-270 -- the array object handed to numpy.save for a pattern id."""
    k = int(pattern[0])
    t = table if k == 0 else numpy.rot90(table, k)
    return numpy.ascontiguousarray(t) if pattern.endswith('.1') else numpy.asfortranarray(t)


def write_pattern(table, pattern):
    return npy_bytes(pattern_array(table, pattern))


def argsort_rows(table):
    """uq.py:773-775 -- argsort of the rows viewed as 'V<C>' (memcmp order).  Stable (Q17)."""
    table = numpy.ascontiguousarray(table)
    keys = [table[:, c] for c in range(table.shape[1] - 1, -1, -1)]
    return numpy.lexsort(keys).astype(numpy.int64)


def unique_rows(table):
    """uq.py:784-789 -- numpy.unique(rows as void, return_inverse=True): distinct rows in
    memcmp order + for each input row the index of its distinct row."""
    order = argsort_rows(table)
    s = table[order]
    n = len(s)
    flag = numpy.ones(n, dtype=bool)
    if n > 1:
        flag[1:] = (s[1:] != s[:-1]).any(axis=1)
    group = numpy.cumsum(flag) - 1
    key = numpy.empty(n, dtype=numpy.int64)
    key[order] = group
    return s[flag], key


def narrow_key(key):
    """uq.py:790 / 832 -- key.astype(numpy.min_scalar_type(max(key)))."""
    return key.astype(numpy.min_scalar_type(int(key.max())))


def encode_dna_qual(table, sort_order, table_name, raw, pattern):
    """uq.py:765-805.  `sort_order`: None = no sort, False = compute and return, ndarray = apply.
    Returns (members {name: npy bytes}, sort_order)."""
    members = {}
    if raw:                                                                              # uq.py:767-781
        out_name = table_name + '.raw'
        if sort_order is None:
            members[out_name] = write_pattern(table, pattern)
        else:
            if sort_order is False:
                sort_order = argsort_rows(table)
            members[out_name] = write_pattern(table[sort_order], pattern)
    else:                                                                                # uq.py:782-802
        out_name = table_name + '.key'
        uniq, key = unique_rows(table)
        key = narrow_key(key)
        if sort_order is None:
            members[out_name] = npy_bytes(key)
        else:
            if sort_order is False:
                sort_order = numpy.argsort(key, kind='stable').astype(numpy.int64)
            members[out_name] = npy_bytes(key[sort_order])
        members[table_name] = write_pattern(uniq, pattern)
    return members, sort_order


def _stack_columns(columns_data):
    """uq.py:814 / 828 -- numpy.dstack(cols)[0]: N x ncols in the widest column dtype."""
    return numpy.dstack(columns_data)[0]


def encode_qname(columns_data, columns, sort_order, raw):
    """uq.py:808-851.  Same tri-state `sort_order`.  Returns (members, sort_order or None)."""
    members = {}
    if raw:
        if sort_order is False:                                                          # uq.py:813-818
            stacked = _stack_columns(columns_data)
            keys = [stacked[:, c] for c in range(stacked.shape[1] - 1, -1, -1)]
            sort_order = numpy.lexsort(keys).astype(numpy.int64)
        if isinstance(sort_order, numpy.ndarray):
            for idx, column in enumerate(columns):
                members[column['name'] + '.raw'] = npy_bytes(columns_data[idx][sort_order])
        else:
            for idx, column in enumerate(columns):
                members[column['name'] + '.raw'] = npy_bytes(columns_data[idx])
    else:                                                                                # uq.py:827-849
        stacked = _stack_columns(columns_data)
        keys = [stacked[:, c] for c in range(stacked.shape[1] - 1, -1, -1)]
        order = numpy.lexsort(keys)
        s = stacked[order]
        n = len(s)
        flag = numpy.ones(n, dtype=bool)
        if n > 1:
            flag[1:] = (s[1:] != s[:-1]).any(axis=1)
        group = numpy.cumsum(flag) - 1
        columns_key = numpy.empty(n, dtype=numpy.int64)
        columns_key[order] = group
        uniq = s[flag]
        columns_key = narrow_key(columns_key)
        if sort_order is False:
            sort_order = numpy.argsort(columns_key, kind='stable').astype(numpy.int64)
        if isinstance(sort_order, numpy.ndarray):
            members['QNAME.key'] = npy_bytes(columns_key[sort_order])
        else:
            members['QNAME.key'] = npy_bytes(columns_key)
        for idx, column in enumerate(columns):
            members[column['name']] = npy_bytes(uniq[:, idx].astype(column['dtype']))
    return members, (sort_order if isinstance(sort_order, numpy.ndarray) else None)


def run_mix(sorted_on, raw_tables, dna, qual, columns_data, columns, pattern):
    """uq.py:739-753 -- the order of the three table builds and who produces `sort_order`."""
    members = {}
    if sorted_on in ['DNA', 'QUAL']:
        not_sorted_on = 'DNA' if sorted_on == 'QUAL' else 'QUAL'
        tabs = {'DNA': dna, 'QUAL': qual}
        pats = {'DNA': pattern[0], 'QUAL': pattern[1]}
        m, sort_order = encode_dna_qual(tabs[sorted_on], False, sorted_on, sorted_on in raw_tables, pats[sorted_on])
        members.update(m)
        m, _ = encode_dna_qual(tabs[not_sorted_on], sort_order, not_sorted_on, not_sorted_on in raw_tables, pats[not_sorted_on])
        members.update(m)
        m, _ = encode_qname(columns_data, columns, sort_order, 'QNAME' in raw_tables)
        members.update(m)
    elif sorted_on == 'QNAME':
        m, sort_order = encode_qname(columns_data, columns, False, 'QNAME' in raw_tables)
        members.update(m)
        m, _ = encode_dna_qual(dna, sort_order, 'DNA', 'DNA' in raw_tables, pattern[0]); members.update(m)
        m, _ = encode_dna_qual(qual, sort_order, 'QUAL', 'QUAL' in raw_tables, pattern[1]); members.update(m)
    else:
        m, _ = encode_qname(columns_data, columns, None, 'QNAME' in raw_tables); members.update(m)
        m, _ = encode_dna_qual(dna, None, 'DNA', 'DNA' in raw_tables, pattern[0]); members.update(m)
        m, _ = encode_dna_qual(qual, None, 'QUAL', 'QUAL' in raw_tables, pattern[1]); members.update(m)
    return members


# --------------------------------------------------------------------------- whole encode
def encode(data, sort=None, raw=None, pattern=None, notricks=False, pad=False):
    """End-to-end restatement of the encode branch (uq.py:73-919) minus printing, --test and tar.
    `sort` in {None,'DNA','QUAL','QNAME'}; `raw` an iterable of table names; `pattern` 2 ids.
    Returns (config dict, members {name: npy bytes}, tables dict)."""
    lines = read_lines(data)
    p1 = pass1(lines)
    d = decide(p1['static_qualities'], p1['dna_min'], p1['dna_max'], notricks, pad)
    columns = qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
    dna, qual = encoder(lines, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'],
                        d['quality_bytes_per_row'], d['bits_per_base'], d['bits_per_quality'],
                        d['variable_read_lengths'])
    columns_data = qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], columns)
    raw_tables = tuple(raw) if raw else (None,)
    if pattern is None: pattern = ['0.1', '0.1']                                         # uq.py:258
    members = run_mix(sort, raw_tables, dna, qual, columns_data, columns, pattern)
    config = {                                                                           # uq.py:681-696, 898-900
        'base_distribution': dict(d['base_graph']), 'qual_distribution': dict(d['qual_graph']),
        'reads': p1['reads'], 'bases': d['bases'], 'qualities': d['qualities'],
        'variable_read_lengths': d['variable_read_lengths'], 'bits_per_base': d['bits_per_base'],
        'bits_per_quality': d['bits_per_quality'], 'N_qual': d['N_qual'], 'dna_max': d['dna_max'],
        'QNAME_prefix': p1['prefix'], 'QNAME_suffix': p1['suffix'], 'QNAME_separators': p1['separators'],
        'QNAME_columns': columns, 'sort': sort if sort else [None], 'raw': list(raw_tables),
        'pattern': list(pattern),
    }
    tables = {'DNA': dna, 'QUAL': qual, 'QNAME': columns_data}
    return config, members, tables


# --------------------------------------------------------------------------- decode
def unpattern(member_bytes, pattern='0.1'):
    """uq.py:943-945 -- numpy.load honours fortran_order; rot90 by -k undoes the rotation."""
    a = numpy.load(io.BytesIO(member_bytes))
    if pattern.startswith('0.'): return a
    return numpy.rot90(a, -int(pattern[0]))


def split_bits(table, total_bits, bits_per_x):
    """uq.py:1002-1007 -- each row as a big-endian integer cut into symbols MSB first."""
    bitmask = (1 << bits_per_x) - 1
    static = range(total_bits - bits_per_x, -bits_per_x, -bits_per_x)
    for row in table:
        the_number = int.from_bytes(bytes(bytearray(row.tolist())), 'big')
        yield [(the_number >> x) & bitmask for x in static]


def decode_tables(config, members):
    """uq.py:951-973 -- tables back to one row per read, QNAME members ordered numerically (Q6)."""
    pat = config['pattern']
    if 'DNA.raw' in members: DNA = unpattern(members['DNA.raw'], pat[0])
    else: DNA = unpattern(members['DNA'], pat[0])[unpattern(members['DNA.key'])]
    if 'QUAL.raw' in members: QUAL = unpattern(members['QUAL.raw'], pat[1])
    else: QUAL = unpattern(members['QUAL'], pat[1])[unpattern(members['QUAL.key'])]
    ncols = len(config['QNAME_columns'])
    if 'QNAME.key' in members:
        cols = [unpattern(members['QNAME_%d' % (i + 1)]) for i in range(ncols)]
        key = unpattern(members['QNAME.key'])
        cols = [c[key] for c in cols]
    else:
        cols = [unpattern(members['QNAME_%d.raw' % (i + 1)]) for i in range(ncols)]
    return DNA, QUAL, cols


def decode(config, members):
    """uq.py:986-1058 -- FASTQ text.  Works on codes (Q25): the sentinel is found as the first
    symbol with code 1 scanning from the MSB, before any character mapping."""
    DNA, QUAL, cols = decode_tables(config, members)
    bases = config['bases']; qualities = config['qualities']
    N_qual = config['N_qual']
    qual_N = dict((v, k) for k, v in N_qual.items())
    variable = config['variable_read_lengths']
    dna_max = config['dna_max']
    bpb = config['bits_per_base']; bpq = config['bits_per_quality']
    total_dna_bits = (variable + dna_max) * bpb
    total_qual_bits = (variable + dna_max) * bpq
    columnType = config['QNAME_columns']
    separators = config['QNAME_separators']
    out = []
    gen = zip(split_bits(DNA, total_dna_bits, bpb), split_bits(QUAL, total_qual_bits, bpq))
    for r, (dna, qual) in enumerate(gen):
        if variable:                                                                     # uq.py:1039-1041 (on codes)
            dna = dna[1 + dna.index(1):]
            qual = qual[1 + qual.index(1):]
        d = []; q = []
        for cd, cq in zip(dna, qual):                                                    # uq.py:1034-1037
            if cq in qual_N:
                d.append(qual_N[cq])
            else:
                d.append(bases[cd])
            q.append(qualities[cq] if cq < len(qualities) else '?')
        name = config['QNAME_prefix']                                                    # uq.py:1010-1024
        for idx, column in enumerate(columnType):
            v = int(cols[idx][r])
            if column['format'] == 'mapping': name += column['map'][v]
            else: name += str(v + column['min']) if column['offset'] else str(v)
            if idx < len(separators): name += separators[idx]
        name += config['QNAME_suffix']
        out.append(name + '\n' + ''.join(d) + '\n+\n' + ''.join(q) + '\n')
    return ''.join(out)


# --------------------------------------------------------------------------- container
def write_tar(path, config, members):
    """uq.py:897-913 -- config.json + members (no '.npy' suffix) in an uncompressed tar."""
    with tarfile.open(path, mode='w') as t:
        blob = json.dumps(config, indent=4, sort_keys=True).encode()
        names = ['config.json'] + sorted(members)
        for name in names:
            data = blob if name == 'config.json' else members[name]
            ti = tarfile.TarInfo(name); ti.size = len(data)
            t.addfile(ti, io.BytesIO(data))


def read_tar(path):
    with tarfile.open(path) as t:
        members = {m.name: t.extractfile(m).read() for m in t.getmembers()}
    config = json.loads(members.pop('config.json').decode())
    return config, members
