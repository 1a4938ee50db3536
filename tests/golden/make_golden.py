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


def make_case(name, fastq_bytes, flags, stable=False, decode=True, outdir=HERE):
    work = tempfile.mkdtemp(prefix='uq_golden_')
    try:
        inp = os.path.join(work, 'in.fastq')
        with open(inp, 'wb') as f: f.write(fastq_bytes)
        tmpd = os.path.join(work, 'tmp'); os.mkdir(tmpd)
        out = os.path.join(work, 'out.uQ')
        rc, log = run_reference(['-i', inp, '-o', out, '--temp', tmpd] + flags, work, stable)
        if rc != 0 or not os.path.isfile(out):
            raise RuntimeError('reference failed for %s:\n%s' % (name, log[-3000:]))
        meta = {'name': name, 'flags': flags, 'stable_patch': stable,
                'input_sha256': hashlib.sha256(fastq_bytes).hexdigest(),
                'reference_sha256': hashlib.sha256(open(REFERENCE, 'rb').read()).hexdigest(),
                'generator': 'tests/golden/make_golden.py'}
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


def cases():
    sys.path.insert(0, REPO)
    from uq_amd import synth
    S = 20261003
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


if __name__ == '__main__':
    only = set(a for a in sys.argv[1:] if not a.startswith('--'))
    outdir = HERE
    for a in sys.argv[1:]:
        if a.startswith('--outdir='): outdir = a.split('=', 1)[1]
    for name, fq, flags, stable in cases():
        if only and name not in only: continue
        make_case(name, fq, flags, stable, outdir=outdir)
