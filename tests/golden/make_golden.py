#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs in the build container only (needs /root/reference/uq.py); the fixtures it writes
are committed, this script is their provenance.  Nothing here is imported by the product.

The reference is a Python-2 script that imports `cffi` (absent from this image) and runs
at import, so it is executed as a derived program, in a scratch directory that is deleted
afterwards:
  1. stdlib lib2to3 translates the text of /root/reference/uq.py in memory;
  2. a few one-line regex patches make it run on py3 / numpy 2 (integer division, text-mode
     open, ravel() of unique's inverse, BytesIO for tar members) -- listed in PATCHES, each
     asserted to hit the expected number of sites;
  3. a memory-only `cffi` shim (malloc / free / cast / new / buffer on ctypes memory, with
     cffi's 0..255 range check on uint8 stores) stands in for the missing package.  The shim
     does no arithmetic: every packed byte is computed by the reference's own loops.
Outputs: for each case `<name>.fastq` (input), `<name>.uQ` (the tar the reference wrote) and
`<name>.json` (argv + provenance + what the reference's own decoder made of the file) and, when that decoder
finished, its output `<name>.refdecode.fastq`.  `--stable` cases add the documented Q17 patch
(argsort kind='stable') so that tie order is comparable.
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference/uq.py'

PATCHES = [
    (r'self\.total /= 4', 'self.total //= 4', 1),
    (r'entries_read/10', 'entries_read//10', 1),
    (r"open\(file_path,\s*'rb'\)", "open(file_path,'r',encoding='latin-1',newline='\\\\n')", 2),
    (r"open\(args\.input,\s*'rb'\)", "open(args.input,'r',encoding='latin-1',newline='\\\\n')", 1),
    (r"open\(inFile,\s*'rb'\)", "open(inFile,'r',encoding='latin-1',newline='\\\\n')", 1),
    (r"with open\(path,\s*'wb'\) as f: f\.write\(json", "with open(path,'w') as f: f.write(json", 1),
    (r'(table,key = numpy\.unique\(table, return_inverse=True\))', r'\1; key = key.ravel()', 1),
    (r'(common_dtype_columns_data, columns_key = numpy\.unique\(common_dtype_columns_data, return_inverse=True\))',
     r'\1; columns_key = columns_key.ravel()', 1),
    (r'numpy\.load\(uq\.extractfile\(file_name\)\)', 'numpy.load(io.BytesIO(uq.extractfile(file_name).read()))', 2),
]
STABLE_PATCHES = [
    (r'numpy\.argsort\(table,axis=0\)', "numpy.argsort(table,axis=0,kind='stable')", 1),
    (r'numpy\.argsort\(key\)', "numpy.argsort(key,kind='stable')", 1),
    (r'numpy\.argsort\(common_dtype_columns_data,axis=0\)', "numpy.argsort(common_dtype_columns_data,axis=0,kind='stable')", 1),
    (r'numpy\.argsort\(columns_key\)', "numpy.argsort(columns_key,kind='stable')", 1),
]

CFFI_SHIM = r'''
import ctypes, re
class _Lib:
    def malloc(self, n):
        buf = (ctypes.c_uint8 * max(int(n), 1))()      # zero-filled, as large mmap'd mallocs are
        return buf
    def free(self, p):
        return 0
class _Row:
    __slots__ = ('buf', 'off', 'cols')
    def __init__(self, buf, off, cols): self.buf = buf; self.off = off; self.cols = cols
    def __setitem__(self, c, v):
        v = int(v)
        if not 0 <= v <= 255: raise OverflowError("integer %d does not fit 'uint8_t'" % v)
        if not 0 <= c < self.cols: raise IndexError(c)
        self.buf[self.off + c] = v
    def __getitem__(self, c): return self.buf[self.off + c]
class _Arr2D:
    def __init__(self, buf, rows, cols): self.buf = buf; self.rows = rows; self.cols = cols
    def __getitem__(self, r): return _Row(self.buf, r * self.cols, self.cols)
class _Arr1D:
    def __init__(self, ctype, n): self.buf = (ctype * n)(); self.ctype = ctype
    def __setitem__(self, i, v):
        v = int(v)
        lim = 1 << (8 * ctypes.sizeof(self.ctype))
        if not 0 <= v < lim: raise OverflowError(v)
        self.buf[i] = v
    def __getitem__(self, i): return self.buf[i]
_CT = {'uint8': ctypes.c_uint8, 'uint16': ctypes.c_uint16, 'uint32': ctypes.c_uint32, 'uint64': ctypes.c_uint64}
class FFI:
    def cdef(self, s): pass
    def dlopen(self, name): return _Lib()
    def cast(self, decl, ptr):
        m = re.match(r'uint8_t\[(\d+)\]\[(\d+)\]', decl)
        return _Arr2D(ptr, int(m.group(1)), int(m.group(2)))
    def new(self, decl):
        m = re.match(r'(\w+)_t\[(\d+)\]', decl)
        return _Arr1D(_CT[m.group(1)], int(m.group(2)))
    def buffer(self, a): return memoryview(a.buf).cast('B')
'''


def translate(stable):
    from lib2to3 import refactor
    src = open(REFERENCE, encoding='latin-1').read()
    fixers = refactor.get_fixers_from_package('lib2to3.fixes')
    tool = refactor.RefactoringTool(fixers)
    out = str(tool.refactor_string(src, 'uq.py'))
    out = 'import io\n' + out
    for pat, rep, n in PATCHES + (STABLE_PATCHES if stable else []):
        out, k = re.subn(pat, rep, out)
        assert k == n, (pat, k, n)
    return out


def run_reference(argv, workdir, stable=False):
    """Run the derived reference with `argv` inside `workdir`; returns its stdout."""
    prog = os.path.join(workdir, '_derived_reference.py')
    with open(prog, 'w') as f: f.write(translate(stable))
    with open(os.path.join(workdir, 'cffi.py'), 'w') as f: f.write(CFFI_SHIM)
    env = dict(os.environ, PYTHONPATH=workdir, PYTHONDONTWRITEBYTECODE='1')
    p = subprocess.run([sys.executable, prog] + argv, cwd=workdir, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return p.returncode, p.stdout


def make_case(name, fastq_bytes, flags, stable=False, decode=True, outdir=HERE, refuses=False):
    work = tempfile.mkdtemp(prefix='uq_golden_')
    try:
        inp = os.path.join(work, 'in.fastq')
        with open(inp, 'wb') as f: f.write(fastq_bytes)
        tmpd = os.path.join(work, 'tmp'); os.mkdir(tmpd)
        out = os.path.join(work, 'out.uQ')
        rc, log = run_reference(['-i', inp, '-o', out, '--temp', tmpd] + flags, work, stable)
        refused = rc != 0 or not os.path.isfile(out)
        if refused and not refuses:
            raise RuntimeError('reference failed for %s:\n%s' % (name, log[-3000:]))
        if refuses and not refused:
            raise RuntimeError('the reference was expected to refuse %s and wrote a file' % name)
        meta = {'name': name, 'flags': flags, 'stable_patch': stable,
                'input_sha256': hashlib.sha256(fastq_bytes).hexdigest(),
                'reference_sha256': hashlib.sha256(open(REFERENCE, 'rb').read()).hexdigest(),
                'generator': 'tests/golden/make_golden.py'}
        if refused:
            # the fixture IS the refusal: what the reference printed last before it gave up (its exit message, or the
            # exception that ended it), no .uQ
            says = [l for l in log.strip().splitlines() if l.strip()][-1]
            meta['reference_refuses'] = True
            meta['reference_says'] = says.strip()[:200]
            with open(os.path.join(outdir, name + '.fastq'), 'wb') as f: f.write(fastq_bytes)
            with open(os.path.join(outdir, name + '.json'), 'w') as f: json.dump(meta, f, indent=1, sort_keys=True)
            print('wrote', name, 'REFUSED:', meta['reference_says'])
            return
        if decode:
            # the reference's own decoder on the file the reference wrote: its stdout is kept as <name>.refdecode.fastq
            # (data: the decoder's output) so that lines 2 and 4 of every record pin our decoder to it
            rc, text = run_reference(['-i', out, '--decode'], work, stable)
            got = text.encode('latin-1')
            meta['reference_decode_roundtrip'] = (rc == 0 and got == fastq_bytes)
            meta['reference_decode'] = describe_decode(rc, got, fastq_bytes, flags)
            ref_out = os.path.join(outdir, name + '.refdecode.fastq')
            if meta['reference_decode']['seq_qual_lines'] != 'unavailable':
                with open(ref_out, 'wb') as f: f.write(got)
            elif os.path.exists(ref_out):
                os.remove(ref_out)
        with open(os.path.join(outdir, name + '.fastq'), 'wb') as f: f.write(fastq_bytes)
        shutil.copyfile(out, os.path.join(outdir, name + '.uQ'))
        with open(os.path.join(outdir, name + '.json'), 'w') as f: json.dump(meta, f, indent=1, sort_keys=True)
        print('wrote', name, os.path.getsize(out), 'bytes', meta.get('reference_decode_roundtrip'))
    finally:
        shutil.rmtree(work, ignore_errors=True)


def describe_decode(rc, got, fastq_bytes, flags):
    """Why `reference_decode_roundtrip` is what it is: which lines of the reference decoder's output equal the input."""
    want = fastq_bytes.split(b'\n')[:-1]
    have = got.split(b'\n')[:-1]
    if rc != 0 or len(have) != len(want) or len(have) % 4:
        tail = got.decode('latin-1', 'replace').strip().splitlines()[-1:] or ['']
        return {'seq_qual_lines': 'unavailable', 'qname_lines': 'unavailable',
                'why': 'the reference decoder stopped (rc %d): %s' % (rc, tail[0][:200])}
    sorted_run = '--sort' in flags and flags[flags.index('--sort') + 1] != 'None'
    pick = lambda lines, k: lines[k::4]
    def same(k):
        a, b = pick(have, k), pick(want, k)
        return a == b if not sorted_run else sorted(a) == sorted(b)
    recs = lambda lines: sorted(zip(pick(lines, 1), pick(lines, 3)))
    d = {'seq_qual_lines': 'equal to the input' + (' as a multiset of (SEQ, QUAL) pairs (the file is sorted)' if sorted_run else '')
         if (same(1) and same(3) and same(2) and recs(have) == recs(want)) else 'DIFFERENT from the input',
         'qname_lines': 'equal to the input' if same(0) else 'different from the input'}
    if d['qname_lines'] != 'equal to the input':
        d['why'] = ('QNAME fields come back permuted: the reference decoder orders the QNAME columns by tar member order = '
                    'os.listdir order (uq.py:911, 964-972; SURVEY.md App. B Q6).  SEQ and QUAL lines are unaffected.')
    return d


def rename(fastq_bytes, name_of):
    """The same reads under other names: line 1 of read i becomes name_of(i) (bytes, with its '@')."""
    lines = fastq_bytes.split(b'\n')
    for i in range(0, len(lines) - 1, 4):
        lines[i] = name_of(i // 4)
    return b'\n'.join(lines)


def custom(n, length, bases, quals, seed, special=None):
    """A FASTQ of n reads over the given alphabets (numpy's legacy RandomState: the same bytes on every numpy): names @c:<i % 7>:<1000 + i>; variable
    lengths avoid multiples of four (Q7).  `special(i, seq, qual, rng)` may edit a read."""
    import numpy as np
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        l = length if isinstance(length, int) else int(rng.randint(length[0], length[1] + 1))
        if not isinstance(length, int) and l % 4 == 0: l += 1 if l < length[1] else -1
        s = ''.join(bases[int(x)] for x in rng.randint(0, len(bases), size=l))
        q = ''.join(quals[int(x)] for x in rng.randint(0, len(quals), size=l))
        if special: s, q = special(i, s, q, rng)
        out.append('@c:%d:%d\n%s\n+\n%s\n' % (i % 7, 1000 + i, s, q))
    return ''.join(out).encode('latin-1')


def case_refuses(name):
    """Cases named *_refused are inputs the reference gives up on (exit message or exception): the fixture is the refusal."""
    return name.endswith('_refused')


def cases():
    sys.path.insert(0, REPO)
    from uq_amd import synth
    S = 20261003
    RAW = ['--raw', 'DNA', 'QUAL', 'QNAME']
    # 1. config 1 of BASELINE.json: 10k x 100bp, raw, no sort, pattern 0.1 0.1
    yield 'cfg1_10k_100bp', synth.fastq(S + 1, 10000, 100), ['--sort', 'None', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '0.1', '0.1'], False
    # 2. fixed length, N gets a NEW quality code (its one quality is shared) -- Q9 corner, 41+1 quals
    yield 'fixed_n_newcode', synth.fastq(S + 11, 300, 50, n_rate=3, n_qual_exclusive=False), ['--raw', 'DNA', 'QUAL', 'QNAME'], False
    # 3. variable length (L % 4 != 0 only: Q7) with the decodable N-trick
    yield 'variable_ntrick', synth.fastq(S + 12, 400, (37, 75), n_rate=2, skip_len_mod4=True), ['--raw', 'DNA', 'QUAL', 'QNAME'], False
    # 4. same input, --notricks -> ACGNT 3-bit path
    yield 'variable_notricks', synth.fastq(S + 12, 400, (37, 75), n_rate=2, skip_len_mod4=True), ['--notricks', '--raw', 'DNA', 'QUAL', 'QNAME'], False
    # 5. --pad: 3-bit -> 4-bit DNA, 6-bit -> 8-bit QUAL
    yield 'fixed_pad', synth.fastq(S + 13, 256, 61, n_rate=2), ['--notricks', '--pad', '--raw', 'DNA', 'QUAL', 'QNAME'], False
    # 6. --sort DNA keyed (default raw=None) with duplicates, patterns 1.2 / 3.1; unstable + stable
    dup = synth.fastq(S + 14, 600, 40, dup='both', dup_templates=24)
    yield 'sort_dna_keyed', dup, ['--sort', 'DNA', '--pattern', '1.2', '3.1'], False
    yield 'sort_dna_keyed_stable', dup, ['--sort', 'DNA', '--pattern', '1.2', '3.1'], True
    # 7. --sort QUAL all raw, patterns 2.2 2.2 (the README's best layout)
    yield 'sort_qual_raw_stable', dup, ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '2.2', '2.2'], True
    # 8. --sort QNAME, QNAME raw / keyed
    yield 'sort_qname_raw_stable', dup, ['--sort', 'QNAME', '--raw', 'QNAME', '--pattern', '0.2', '1.1'], True
    yield 'sort_qname_keyed_stable', dup, ['--sort', 'QNAME', '--pattern', '2.1', '3.2'], True
    # 9. no sort, everything keyed
    yield 'nosort_keyed', dup, ['--sort', 'None', '--pattern', '1.1', '1.2'], False
    # 10. mixed: sort DNA raw, QUAL keyed
    yield 'sort_dna_raw_qual_keyed_stable', dup, ['--sort', 'DNA', '--raw', 'DNA', '--pattern', '3.1', '0.2'], True
    # 11. the UNPATCHED reference (numpy's default, unstable argsort) on the sorted mixes: compared by the Q17 rule
    #     (sorted-on table / unique tables / sorted-on key exact, the rest a multiset inside every tie group)
    yield 'sort_qual_raw', dup, ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '2.2', '2.2'], False
    yield 'sort_qname_raw', dup, ['--sort', 'QNAME', '--raw', 'QNAME', '--pattern', '0.2', '1.1'], False
    yield 'sort_qname_keyed', dup, ['--sort', 'QNAME', '--pattern', '2.1', '3.2'], False
    yield 'sort_dna_raw_all', dup, ['--sort', 'DNA', '--raw', 'DNA', 'QUAL', 'QNAME'], False

    # ---- round 4: other QNAME grammars (uq.py:394-444, 555-678, 717-736).  Everything above shares ONE grammar
    # (@SIM001:42:FCX01:a:b:x:y -- ':' separators, four integer columns of one or two bytes).
    # 12. Illumina 1.8 names with the comment: a space among the separators, two `mapping` columns (filter flag, barcode)
    BC = [b'ATCACG', b'CGATGT', b'TTAGGC', b'TGACCA']
    illumina = lambda i: b'@M01234:17:000000000-A1B2C:1:%d:%d:%d %d:%s:0:%s' % (
        1101 + 1000 * (i % 2) + i % 19, 1000 + (i * 7919) % 28000, 1000 + (i * 104729) % 28000, 1 + i % 2, b'Y' if i % 10 == 3 else b'N', BC[(i // 3) % 4])
    ill = rename(synth.fastq(S + 20, 500, 36), illumina)
    yield 'qn_illumina_comment', ill, RAW, False
    #     ... and the same names as the sorted-on, keyed table: mapping codes and integers stacked into one table (uq.py:808-851)
    yield 'qn_illumina_sort_qname_keyed_stable', ill, ['--sort', 'QNAME', '--pattern', '1.1', '2.2'], True
    # 13. suffixes: '/1' behind a barcode; a constant ' length=30' whose characters (space, '=') must leave the separator set (uq.py:428-431)
    yield 'qn_suffix_mate', rename(synth.fastq(S + 21, 400, 30), lambda i: b'@ERR0042_%d_%d#ACGT/1' % (i + 1, (5 * i + 3) % 977)), RAW, False
    yield 'qn_suffix_length', rename(synth.fastq(S + 21, 400, 30), lambda i: b'@SRR001666:%d:%d length=30' % (i + 1, (5 * i + 3) % 977)), RAW, False
    # 14. four different separators in one name ('-', '=', ';' after a prefix that ends in '_'), a mapping of strings between integers,
    #     an integer column that needs an offset
    yield 'qn_seps_mixed', rename(synth.fastq(S + 22, 300, 25), lambda i: b'@run7_%d-%d=%s;%d' % (
        i % 50, 100000 + 37 * i, [b'fwd', b'rev', b'unk'][i % 3], i % 2)), RAW, False
    # 15. a separator that is also a minus sign: its count is not constant, the fields around it merge into a column of strings -> the reference exits
    yield 'qn_strings_refused', rename(synth.fastq(S + 22, 300, 25), lambda i: b'@run7_%d-%d_%s' % (
        i % 50 - 25, 100000 + 37 * i, [b'fwd', b'rev'][i % 2])), RAW, False
    # 16. Q14 (the family the CLI fuzz found, fd9f468): a separator candidate that is line 1's last character before the suffix sits in the
    #     separator set but not in the order; what happens depends on the LAST read alone
    q14 = synth.fastq(S + 23, 41, 20)
    yield 'qn_q14_two_refused', rename(synth.fastq(S + 23, 2, 20), lambda i: [b'@q_0=1', b'@q_1=4'][i]), RAW, False
    yield 'qn_q14_three', rename(synth.fastq(S + 23, 3, 20), lambda i: [b'@q_0=1', b'@q_1=4', b'@q_2=1'][i]), RAW, False
    yield 'qn_q14_many', rename(q14, lambda i: (b'@q_%d=1' % (i % 7)) if i < 40 else b'@q_1=4'), RAW, False
    # 17. Q13: no constant-count separator at all / only line 1's last character -> re.error inside order_seps (uq.py:433-438)
    yield 'qn_no_separator_refused', rename(synth.fastq(S + 23, 5, 20), lambda i: b'@r%d' % (i + 1)), RAW, False
    yield 'qn_q14_const_refused', rename(synth.fastq(S + 23, 30, 20), lambda i: b'@r:7:%d:7' % i), RAW, False
    # 18. Q15: '.' goes into the regexes unescaped
    yield 'qn_dot_separator_refused', rename(synth.fastq(S + 23, 5, 20), lambda i: b'@a.%d.%s' % (i + 1, b'xyzuv'[i:i + 1])), RAW, False
    # 19. column widths and offsets (uq.py:641-670): negative values (offset), a uint32 and a uint64 range, a small range far from zero (offset),
    #     four wide-spread numbers that stay a mapping of strings; and the same columns as the sorted-on raw table (one common dtype, uq.py:812-816)
    wide = rename(synth.fastq(S + 24, 600, 24), lambda i: b'@s:%d:%d:%d:%d:%d' % (
        i % 50 - 25, 100000 + 37000 * i, 5000000000 * (i % 300) + 7 * i, 1000 + i % 90, [5, 70000, 12345678, 31][i % 4]))
    yield 'qn_u4_u8_negative', wide, RAW, False
    yield 'qn_u4_u8_sort_qname_raw_stable', wide, ['--sort', 'QNAME', '--raw', 'QNAME', '--pattern', '0.1', '3.2'], True
    # 20. the checkpoints of uq.py:586-602 at 22 000 reads: column 1 is a mapping at read 10 000 (900 values) and integers at 20 000;
    #     column 2 integers at once; column 3 two strings; column 4 three far-apart numbers until read 21 000, then one per read -- still a
    #     mapping at the last check (1 002 values <= 2 199), stored as uint16 codes of strings
    def demote(i):
        a = i % 900 if i < 10000 else i
        d = [12, 999999, 70000][i % 3] if i < 21000 else i
        return b'@d:%d:%d:%s:%d' % (a, 3 * i, [b'L', b'R'][i % 2], d)
    yield 'qn_demote_20000', rename(synth.fastq(S + 25, 22000, 12), demote), RAW, False
    #     a column that became integers at read 10 000 meets a non-number at read 15 000 -> strings -> the reference exits
    yield 'qn_demoted_then_string_refused', rename(synth.fastq(S + 25, 15200, 10),
                                                   lambda i: b'@d:%d:%s' % (i % 5, (b'%d' % i) if i != 15000 else b'x15000')), RAW, False

    # ---- round 4: sort x geometry (uq.py:765-851).  Every sorted fixture above is fixed-length 40 bp with the default tricks.
    # 21. variable lengths (sentinel rows) + --sort DNA keyed; + --sort QUAL raw
    var = synth.fastq(S + 26, 700, (31, 58), n_rate=2, skip_len_mod4=True, dup='both', dup_templates=30)
    yield 'var_sort_dna_keyed_stable', var, ['--sort', 'DNA', '--pattern', '2.2', '1.1'], True
    yield 'var_sort_dna_keyed', var, ['--sort', 'DNA', '--pattern', '2.2', '1.1'], False
    yield 'var_sort_qual_raw_stable', var, ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '3.2', '1.1'], True
    # 22. --pad (4-bit DNA, 8-bit QUAL) + --sort QUAL, QUAL raw and DNA / QNAME keyed
    yield 'pad_sort_qual_stable', synth.fastq(S + 27, 500, 33, n_rate=2, dup='qual', dup_templates=20), \
        ['--notricks', '--pad', '--sort', 'QUAL', '--raw', 'QUAL', '--pattern', '0.2', '3.1'], True
    # 23. --notricks (3-bit ACGNT) keyed, unsorted and sorted, two non-trivial patterns each
    nt = synth.fastq(S + 28, 500, 45, n_rate=3, dup='dna', dup_templates=25)
    yield 'notricks_keyed', nt, ['--notricks', '--sort', 'None', '--pattern', '1.1', '2.2'], False
    yield 'notricks_sort_dna_keyed_stable', nt, ['--notricks', '--sort', 'DNA', '--pattern', '3.2', '2.1'], True

    # ---- round 4: alphabets and widths (uq.py:448-457, 477-545).  The synthetic generator only makes ACGT(N) with 41 qualities: 2- / 3-bit bases, 6-bit
    # qualities.  24. eleven IUPAC bases -> 4-bit DNA, 30 qualities -> 5 bits; the same sorted and keyed
    iupac = custom(300, 33, 'ACGTNRYKMSW', [chr(33 + k) for k in range(30)], 1)
    yield 'alpha_iupac_4bit', iupac, RAW, False
    yield 'alpha_iupac_sort_dna_keyed_stable', custom(300, 33, 'ACGTNRYKMSW', [chr(33 + k) for k in range(30)], 8), ['--sort', 'DNA', '--pattern', '1.1', '0.2'], True
    # 25. 70 qualities -> 7 bits; 133 qualities (bytes beyond 127 among them) -> 8 bits
    yield 'qual_7bit', custom(300, 29, 'ACGT', [chr(33 + k) for k in range(70)], 2), RAW, False
    yield 'qual_8bit', custom(300, 21, 'ACGT', [chr(33 + k) for k in range(93)] + [chr(161 + k) for k in range(40)], 3), RAW, False
    # 26. TWO N-trick bases (uq.py:480-494): N always with '!' and nothing else has '!' (its code is that quality's index), X always with '#' which
    #     other bases share (a NEW quality code, Q9); the order of the two is the order of first appearance (Q11)
    def two_ntrick(i, s, q, rng):
        s, q = list(s), list(q)
        for k in range(len(s)):
            if rng.rand() < 0.03: s[k] = 'N'; q[k] = '!'
            elif rng.rand() < 0.03: s[k] = 'X'; q[k] = '#'
            elif q[k] == '!': q[k] = '$'
        return ''.join(s), ''.join(q)
    yield 'two_ntrick_bases', custom(400, 41, 'ACGT', [chr(35 + k) for k in range(20)], 4, two_ntrick), RAW, False
    # 27. one base and one quality (2 bits each, the ladder's floor); upper and lower case (eight bases -> 3 bits); three bases and three qualities, variable lengths
    yield 'one_base_one_qual', custom(50, 10, 'A', 'I', 5), RAW, False
    yield 'alpha_mixed_case', custom(300, 37, 'ACGTacgt', [chr(40 + k) for k in range(12)], 6), RAW, False
    yield 'var_tiny_alphabets', custom(300, (5, 23), 'ACG', '#5I', 7), RAW, False


if __name__ == '__main__':
    only = set(a for a in sys.argv[1:] if not a.startswith('--'))
    outdir = HERE
    for a in sys.argv[1:]:
        if a.startswith('--outdir='): outdir = a.split('=', 1)[1]
    for name, fq, flags, stable in cases():
        if only and name not in only: continue
        make_case(name, fq, flags, stable, outdir=outdir, refuses=case_refuses(name))
