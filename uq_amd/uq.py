#!/usr/bin/env python3
"""uq -- FASTQ <-> uQ (tar of .npy arrays + config.json) on MI355X.  Drop-in for the reference script:

    python -m uq_amd.uq -i reads.fastq [--sort DNA|QUAL|QNAME|None] [--raw DNA QUAL QNAME] [--pattern a b]
                        [--notricks] [--pad] [--peek] [--test [--compressor CMD]] [-o out.uQ] [--temp DIR]
    python -m uq_amd.uq -i reads.fastq.uQ --decode > reads.fastq

Same thirteen flags and validation as uq.py:21-71, same container (uq.py:897-913), same function names
for the seams of the hot path -- `encoder_fixed`, `encoder_variable`, `write_pattern`, `write_out`,
`test_patterns`, `run_mix`, `encode_dna_qual`, `encode_qname`, `load_from_tar`, `split_bits` -- as
methods of `Session` (the reference keeps their shared state in module globals).  Every O(N*L) step
runs on the GPU through the C ABI (uq_amd._lib); there is no CPU fallback.  Between the passes the
tables stay in HBM (the reference parks them in <temp>/*.temp files to save RAM, uq.py:710-711, 734).

Deliberate fixes of reference defects: Q5 private temp dir, Q6 members written
and decoded in numeric order, Q4 compressors spawned on demand, Q13/Q21 clear errors, Q17 stable ties,
Q19/Q25 decoder works on codes.
"""
import argparse
import io
import json
import os
import shutil
import subprocess
import sys
import tarfile
import tempfile
import time

import numpy as np

PATTERNS = ['0.1', '1.1', '2.1', '3.1', '0.2', '1.2', '2.2', '3.2']


class UqError(Exception):
    """The reference prints and exit()s (uq.py:48-50); the library form raises, main() prints."""


def error(message):
    raise UqError(message)


def build_parser():
    p = argparse.ArgumentParser(description='This tool converys FASTQ files to microq (uQ) files and back.')
    p.add_argument('-i', '--input', required=True, help='Required. Input FASTQ/uQ file path.')
    p.add_argument('-o', '--output', help='Optional. FASTQ->uQ only. Default is to append .uQ to input filename.')
    p.add_argument('--compressor', action='store', help='Optional. Path to compression program that accepts data on stdin and prints to stdout/pipe)')
    p.add_argument('--sort', action='store', help='Optional. [DNA/QUAL/QNAME/None] Resort FASTQ. See output of --test for optimium method.')
    p.add_argument('--raw', nargs='+', metavar='file name', help='Optional. [DNA/QUAL/QNAME/None] Store tables raw rather than unique & sorted + key.')
    p.add_argument('--pattern', nargs='+', metavar='pattern', help='Optional. [0.1/0.2/1.1/1.2/2.1/2.2/3.1/3.2] x2 (DNA|QUAL) See output of --test for optimium.')
    p.add_argument('--temp', help='Optional. Directory to write temporary files to. Default is OS dependant.')
    p.add_argument('--test', action='store_true', default=False, help='Optional. Try all possible sort/raw/pattern combinations.')
    p.add_argument('--notricks', action='store_true', default=False, help='Optional. Prevents conversion of N to the most popular base.')
    p.add_argument('--pad', action='store_true', default=False, help='Optional. Pads DNA/QUAL to the nearest 2/4/8 bits.')
    p.add_argument('--peek', action='store_true', default=False, help='Optional. No output files are created, the input is just scanned.')
    p.add_argument('--decode', action='store_true', default=False, help='Requred if you want to convert a .uQ back to .fastq')
    p.add_argument('--device', type=int, default=int(os.environ.get('LOCAL_RANK', '0')), help='GPU index (extension)')
    p.add_argument('--quiet', action='store_true', default=False, help='suppress the analysis report (extension)')
    p.add_argument('--host-qname', action='store_true', default=False, help='run the QNAME passes sequentially on the host (extension)')
    p.add_argument('--exact-qname', action='store_true', default=False,
                   help='run the QNAME passes as kernels of their own (layout, then tokeniser) instead of inside the pack kernel (extension)')
    p.add_argument('--multi-pass', action='store_true', default=False,
                   help='census, record index, statistics and pack as separate passes over the stream; the default counts the statistics in the pack kernel, '
                        'packing speculatively with decisions guessed from the head of the file and verified afterwards (extension)')
    p.add_argument('--two-pass-decode', action='store_true', default=False,
                   help='decode through the fixed-pitch text arrays (uq_unpack + uq_emit_fastq) instead of the fused kernel (extension)')
    return p


def validate_args(args):
    """uq.py:52-71, including the case quirks of Q1."""
    if args.pattern:
        if len(args.pattern) != 2: error('ERROR: There must be 2 values for --pattern!')
        elif not all(p in PATTERNS for p in args.pattern): error('ERROR: Pattern values are incorrect!')
    if args.sort:
        if args.sort.lower() not in ['dna', 'qual', 'qname', 'none']: error('ERROR: --sort value is incorrect!')
        if args.sort.lower() == 'none': args.sort = (None,)
    if args.raw:
        if not all(r.lower() in ['dna', 'qual', 'qname', 'none'] for r in args.raw): error('ERROR: --raw values are incorrect!')
        args.raw = set(args.raw)
        if 'none' in args.raw:
            args.raw.add(None); args.raw.discard('none')
    if not os.path.isfile(args.input): error('ERROR: Sorry, the input path you have specified is not a file!')
    return args


def npy_header(shape, fortran_order, dtype):
    """The .npy v1.0 header numpy.save writes for this array (uq.py:263-274 via numpy)."""
    f = io.BytesIO()
    np.lib.format.write_array_header_1_0(f, {'descr': np.lib.format.dtype_to_descr(np.dtype(dtype)),
                                             'fortran_order': bool(fortran_order), 'shape': tuple(int(s) for s in shape)})
    return f.getvalue()


def pattern_header(rows, cols, pattern):
    """Shape / fortran_order of numpy.save(asXarray(rot90(table, k))) -- SURVEY.md A.4, Q22."""
    k = int(pattern[0])
    shape = (rows, cols) if k % 2 == 0 else (cols, rows)
    fortran = pattern.endswith('.2') and min(shape) > 1     # 1-wide arrays are C- and F-contiguous: numpy says False
    return npy_header(shape, fortran, np.uint8)


class Session:
    """One encode (or decode) run: the reference's module-level state as an object."""

    def __init__(self, args, ctx=None, out=sys.stdout):
        from . import ops
        from .device import Context
        self.ops = ops
        self.args = args
        self.ctx = ctx or Context(getattr(args, 'device', 0))
        from .hostio import Staging
        self.io = Staging(self.ctx)
        self.out = out
        self.members = {}          # name -> (npy header bytes, payload: device tensor or numpy array), uq.py:272-274
        self.tables = {}           # 'DNA' / 'QUAL' -> (device tensor, rows, cols); 'QNAME' -> [device column tensors]
        self.columns = []
        self.config = {}
        self.t0 = time.time()
        self.split = self.t0
        self.last_subprocess_used = 0

    def say(self, *a):
        if not getattr(self.args, 'quiet', False):
            print(*a, file=self.out)

    def split_time(self):
        now = time.time(); d = now - self.split; self.split = now
        return '(' + str(d / 60) + ' minutes)'

    last_params = None         # uq_pack_params of the previous encode in this process

    # ------------------------------------------------------------------ passes 1-2 (analysis)
    def load(self, path):
        """Read the FASTQ, put it in HBM, build the record index.  Replaces `wc -l` + line iteration."""
        ops, ctx = self.ops, self.ctx
        if os.path.getsize(path) == 0: error('ERROR: empty input')
        self.path, self._host = path, None
        # pinned, chunked, file reads overlapped with the PCIe copies; the newline census of a chunk is queued right behind its
        # copy (row f2), so that only the scan of the per-tile counts is left when the last byte lands
        census = {}

        def on_chunk(d, lo, n):
            if 'c' not in census: census['c'] = ops.ChunkedCensus(ctx, d)
            census['c'].chunk(lo, n)
        chunked = self.io.chunk % (16 << 10) == 0
        d_buf = self.io.file_to_device(path, on_chunk=on_chunk if chunked else None)
        self.load_device(d_buf, census=census.get('c'))

    def load_device(self, d_buf, census=None):
        """The same for FASTQ bytes that are already in HBM (a uint8 device tensor; `census`: an ops.ChunkedCensus of it whose chunks have
        all been queued -- load() queues them behind the PCIe copies).  The default is the step bench.py times, queued back
        to back: the census's closing scan (the line count stays on the device) -> the QNAME layout guess from a sample of the reads ->
        pack + pass-1 statistics + QNAME fields in ONE kernel, with decisions guessed from the head of the file (a second context's
        stream works them out beside the census) and verified afterwards against the whole file's counts.  NO record index is written:
        the kernels walk the census's newline lists; `d_ls` is expanded on first use by whatever still wants it (the exact QNAME kernels,
        the plain packers, the N-trick's first occurrences).  The stream is read twice.  --multi-pass: census -> index -> statistics ->
        pack as separate passes (three reads).  Same results whichever path."""
        ops, ctx, args = self.ops, self.ctx, self.args
        self.d_buf = d_buf
        if not hasattr(self, '_host'): self.path, self._host = None, None
        self._spec, self._fq, self._d_ls, self.load_path = None, None, None, 'multi-pass'
        self._guess_shared = False
        nbytes = int(d_buf.numel())
        nlines, queued, guess, cap = None, None, None, 0
        if not getattr(args, 'multi_pass', False) and nbytes:
            if census is None:
                ctx.sync()                               # (whoever made the buffer may still be writing it: the guess's stream does not wait for this one)
                census = ops.ChunkedCensus(ctx, d_buf)
                census.chunk(0, nbytes)
            census.end_async()
            if getattr(self, 'side', None) is None:
                from .device import SideContext
                self.side = SideContext(ctx)            # the guess's small census / index / statistics must not touch the queued census's state
            g = ops.head_guess(self.side, d_buf, notricks=args.notricks, pad=args.pad, head_bytes=ops.HEAD_BYTES_SMALL, head_reads=ops.HEAD_READS_INDEXED)
            if g is not None:
                guess, rpb = g
                cap = int(nbytes * rpb * 1.02) + 1024
                guess.avg_record_bytes = int(1.0 / rpb)
                fq = None
                if self.fused_qname_enabled():
                    fq = ops.FusedQname(ctx, cap)
                    if self.owns_qname_guess(): ops.qname_guess_async(ctx, d_buf, None, fq)
                    self.share_qname_guess(fq)
                queued = ops.pack_stats_async(ctx, d_buf, None, cap, guess, fq=fq)
                if queued is not None and fq is not None: ops.qname_fused_finish(ctx, fq)
            nlines, ok = census.wait()
            if not ok:                                   # a tile with more newlines than a list holds (lines of a few bytes): the bitmap form
                nlines, queued = ops.count_lines(ctx, d_buf), None
        elif census is not None:
            nlines = census.end()
        if nlines is None:
            nlines = ops.count_lines(ctx, self.d_buf)
        if nlines % 4 != 0:
            error('ERROR: The FASTQ file provided contains' + str(nlines) + 'rows, which is not divisible by 4!')
        if nlines == 0: error('ERROR: empty input')
        self.total = nlines // 4
        if queued is not None and self.total <= cap:
            n = self.total
            self._spec = (guess, queued[0][:n * guess.dna_bytes_per_row], queued[1][:n * guess.quality_bytes_per_row], queued[2])
            self.d_stats = queued[3]
            self.load_path = 'two reads (census; pack + statistics), queued, no record index'
            self._fq = fq
        elif not getattr(args, 'multi_pass', False):
            # the queued form does not hold for this file (more reads than the head promised, a head without a whole record, an alphabet
            # without a fused kernel): the same two reads through the plain calls, with the index
            del queued
            guess = ops.head_guess_indexed(ctx, self.d_buf, self.d_ls, self.total, args.notricks, args.pad)
            fq = None
            if guess is not None and self.fused_qname_enabled() and not self._guess_shared:
                fq = ops.FusedQname(ctx, self.total)
                if self.owns_qname_guess(): ops.qname_guess(ctx, self.d_buf, self.d_ls, self.total, fq)
                self.share_qname_guess(fq)
            res = ops.pack_stats(ctx, self.d_buf, self.d_ls, 0, self.total, guess, fq=fq) if guess is not None else None
            if res is not None:
                self._spec = (guess,) + res[:3]
                self.d_stats = res[3]
                self.load_path = 'two reads (census; pack + statistics)'
                if fq is not None:
                    ops.qname_fused_finish(ctx, fq)
                    self._fq = fq
        if self._spec is None:
            self.d_stats = ops.stats_new(ctx)
            ops.stats_accumulate(ctx, self.d_stats, self.d_buf, self.d_ls, 0, self.total)

    # the fused QNAME pass's seams (the sharded session overrides the last two: rank 0's guess is every rank's)
    def fused_qname_enabled(self):
        return not getattr(self.args, 'multi_pass', False) and not getattr(self.args, 'host_qname', False) and not getattr(self.args, 'exact_qname', False)

    def owns_qname_guess(self):
        return True

    def share_qname_guess(self, fq):
        """One GPU: nothing to share (and a fallback may guess again).  The sharded session broadcasts rank 0's structure here -- once per load,
        which is what `_guess_shared` records."""

    @property
    def d_ls(self):
        """The record index (uint64 offset of every line start), expanded when somebody asks for it: the default encode never does."""
        if getattr(self, '_d_ls', None) is None:
            self._d_ls = self.ops.index_lines(self.ctx, self.d_buf, 4 * self.total)
        return self._d_ls

    @d_ls.setter
    def d_ls(self, value):
        self._d_ls = value

    @property
    def host(self):
        """Host copy of the FASTQ bytes: only the sequential QNAME fallback needs it."""
        if self._host is None:
            self._host = np.fromfile(self.path, dtype=np.uint8) if self.path else self.ctx.to_numpy(self.d_buf)
        return self._host

    # small seams the sharded session (uq_amd/dist_encode.py) overrides
    def fetch_stats(self):
        hs = self.ops.stats_fetch(self.ctx, self.d_stats)
        if hs.incomplete:                        # the speculative pass could not count everything: plain statistics pass
            self._spec = None
            self.d_stats = self.ops.stats_new(self.ctx)
            self.ops.stats_accumulate(self.ctx, self.d_stats, self.d_buf, self.d_ls, 0, self.total)
            hs = self.ops.stats_fetch(self.ctx, self.d_stats)
        return hs

    def starts_with_at(self):
        return int(self.d_buf[0]) == ord('@')

    def first_seen(self):
        return self.ops.first_occurrence(self.ctx, self.d_buf, self.d_ls, 0, self.total)

    def reads_in_file(self):
        return self.total

    def analyse_qname(self):
        """QNAME passes 1 / 2 / 4: per-read work on the device (qname_device), or -- for QNAMEs outside the
        subset that path reproduces exactly, and with --host-qname -- sequentially on the host."""
        from . import qname, qname_device
        self.qname_path = 'fused'
        fq, self._fq = getattr(self, '_fq', None), None
        res = qname_device.analyse_fused(self.ctx, fq, self.total) if fq is not None else None
        del fq
        if res is None:
            self.qname_path = 'device'
            res = None if getattr(self.args, 'host_qname', False) else qname_device.analyse_device(self.ctx, self.d_buf, self.d_ls, self.total)
        if res is None:
            self.qname_path = 'host-native'
            h_ls = self.ctx.to_numpy(self.d_ls, np.uint64)
            res = qname.analyse_native(self.host, h_ls, self.total)               # C++ host path
            if res is None:                                                        # regex-special separators etc.: Python path
                self.qname_path = 'host-python'
                res = qname.analyse(qname.qname_lines(self.host, h_ls, self.total))
        return res

    def analyse(self):
        """Pass 1 (histogram, lengths, QNAME layout), the N-trick / width decisions, pass 2 (QNAME typing)."""
        from . import analysis, qname
        args = self.args
        hs = self.fetch_stats()
        if not self.starts_with_at(): error('ERROR: This does not look like a FASTA/FASTQ file! (first line does not start with @)')
        if hs.bad_plus is not None:
            error('ERROR: For entry' + str(hs.bad_plus) + 'the third line does not start with +')
        if hs.bad_len is not None:
            error('ERROR: Length of DNA does not match the length of the quality scores for entry ' + str(hs.bad_len))
        self.hs = hs
        d = analysis.decide_from_stats(hs, notricks=args.notricks, pad=args.pad, first_seen=self.first_seen)
        self.d = d
        try:
            prefix, suffix, separators, columns, arrays = self.analyse_qname()
        except qname.QnameError as e:
            error(str(e))
        self.columns = columns
        self.qname_arrays = arrays
        self.report(prefix, suffix, separators)
        self.config = {
            'base_distribution': d['base_distribution'], 'qual_distribution': d['qual_distribution'],
            'reads': self.reads_in_file(), 'bases': d['bases'], 'qualities': d['qualities'],
            'variable_read_lengths': d['variable_read_lengths'], 'bits_per_base': d['bits_per_base'],
            'bits_per_quality': d['bits_per_quality'], 'N_qual': d['N_qual'], 'dna_max': d['dna_max'],
            'QNAME_prefix': prefix, 'QNAME_suffix': suffix, 'QNAME_separators': separators, 'QNAME_columns': columns,
        }

    def report(self, prefix, suffix, separators):
        """The analysis printout of uq.py:459-545 (condensed: same facts, fewer bar charts)."""
        d, say = self.d, self.say
        total_bases = float(sum(d['base_distribution'].values()))
        say('Finished analysing file!', self.split_time())
        say('    - total reads:', self.reads_in_file())
        say('    - total bases:', int(total_bases))
        say('\nQNAME Analysis:')
        if prefix: say('    - all QNAMEs prefixed with:', prefix)
        if suffix: say('    - all QNAMEs suffixed with:', suffix)
        if separators: say('    - QNAME field separators:', separators, '(', len(separators), 'symbols )')
        say('\nDNA Analysis:')
        bases_all = sorted(d['base_distribution'])
        say('    - DNA sequences contained the following characters:', ' '.join(bases_all), '(' + str(len(bases_all)) + ' in total)')
        for b in bases_all:
            pct = d['base_distribution'][b] / total_bases * 100
            say('     ', b, '|' + ('#' * int(pct / 2)).ljust(50) + '|', ('%.3f' % pct).rjust(7) + '%', str(d['base_distribution'][b]).rjust(12))
        for b, code in d['N_qual'].items():
            say('    - The base', b, 'consistently had the same quality value; it is encoded as', d['bases'][0], 'with quality code', code)
        say('    - this means we will store each letter of DNA in', d['bits_per_base'], 'bits.')
        if d['dna_min'] == d['dna_max']: say('    - all DNA sequences are', d['dna_min'], 'bases long')
        else:
            say('    - the largest DNA sequence is', d['dna_max'], 'bases long')
            say('    - the smallest DNA sequence is', d['dna_min'], 'bases long')
        say('    - we will use', d['dna_bytes_per_row'], 'bytes per unique DNA sequence.\n')
        say('QUAL Analysis:')
        quals_all = sorted(d['qual_distribution'])
        counts = getattr(getattr(self, 'hs', None), 'counts', None)
        if counts is not None:                                                  # uq.py:520-526: the per-base breakdown
            for b in bases_all:
                say('  [Breakdown for "' + b + '"]')
                row = counts[ord(b)]
                for q in np.flatnonzero(row):
                    pct = int(row[q]) / float(d['base_distribution'][b]) * 100
                    say('     ', chr(q), '|' + ('#' * int(pct / 2)).ljust(50) + '|', ('%.3f' % pct).rjust(7) + '%', str(int(row[q])).rjust(12))
                say('')
        say('  [Total distribution]')
        say('    - the following values were seen as quality scores:', ' '.join(quals_all), '(' + str(len(quals_all)) + ' in total)')
        say('    - There distribution is:')
        for q in quals_all:                                                     # uq.py:531-532
            pct = d['qual_distribution'][q] / total_bases * 100
            say('     ', q, '|' + ('#' * int(pct / 2)).ljust(50) + '|', ('%.3f' % pct).rjust(7) + '%', str(d['qual_distribution'][q]).rjust(12))
        nnew = len([c for c in d['N_qual'].values() if c >= len(d['qualities'])])
        if nnew: say('    - as mentioned above,', nnew, 'unique quality scores will be added to the', len(quals_all), 'above.')
        say('    - this means we will store each quality symbol in', d['bits_per_quality'], 'bits.')
        say('    - we will use', d['quality_bytes_per_row'], 'bytes per unique quality sequence.\n')
        for idx, c in enumerate(self.columns):
            say('    - Column', idx + 1, 'is type', c['format'], 'stored as', c['dtype'])

    # ------------------------------------------------------------------ pass 3: the packers
    def _encode(self, variable):
        ops, ctx, d = self.ops, self.ctx, self.d
        p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                                 variable, d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], self.hs.max_record_bytes,
                                 avg_record_bytes=self.d_buf.numel() // max(self.total, 1))
        spec = getattr(self, '_spec', None)
        self._spec = None
        self.pack_path = 'speculative' if (spec is not None and ops.same_pack_params(p, spec[0])) else 'plain'
        if self.pack_path == 'speculative':
            dna, qual, bad = spec[1:]            # packed while the statistics were counted: the guess was right
        else:
            del spec
            dna, qual, bad = ops.pack(ctx, self.d_buf, self.d_ls, 0, self.total, p)
        Session.last_params = p
        b = ops.bad_index(bad) if bad is not None else None      # (the speculative kernel flags its statistics instead)
        if b is not None: error('ERROR: read %d holds a symbol with no code (internal inconsistency)' % b)
        return ((dna, self.total, d['dna_bytes_per_row']), (qual, self.total, d['quality_bytes_per_row']))

    def encoder_fixed(self):
        """uq.py:108-182 on the session's resident buffer.  Returns ((dna tensor, rows, cols), (qual tensor, rows, cols)).
        The module-level `encoder_fixed(...)` / `encoder_variable(...)` keep the reference's ten-argument signature and
        its `((array, ptr), (array, ptr), lib)` return."""
        return self._encode(False)

    def encoder_variable(self):
        """uq.py:188-254 (adds the sentinel bit above each read's most significant symbol)."""
        return self._encode(True)

    def pack(self):
        """uq.py:705-736: pass 3, and pass 4's column arrays moved to the device."""
        arrays = self.encoder_variable() if self.d['variable_read_lengths'] else self.encoder_fixed()
        self.tables['DNA'], self.tables['QUAL'] = arrays
        self.tables['QNAME'] = [a if self.ctx.torch.is_tensor(a) else self.ctx.to_device(a) for a in self.qname_arrays]

    # ------------------------------------------------------------------ writers
    def write_pattern(self, table, filename):
        """uq.py:257-270.  `table` = (device tensor, rows, cols)."""
        args = self.args
        if args.pattern is None: pattern = '0.1'; args.pattern = ['0.1', '0.1']
        elif filename.startswith('DNA'): pattern = args.pattern[0]
        elif filename.startswith('QUAL'): pattern = args.pattern[1]
        else: error('ERROR: This should never happen!')
        t, rows, cols = table
        payload = t[:rows * cols] if pattern == '0.1' else self.ops.pattern(self.ctx, t, rows, cols, pattern)   # 0.1 is the table itself
        self.members[filename] = (pattern_header(rows, cols, pattern), payload)      # stays in HBM until write_container streams it

    def write_out(self, array, filename, dtype=None):
        """uq.py:272-274: a 1-D array (key or QNAME column) as a .npy member.  `array`: device tensor (with the
        numpy dtype its bytes stand for) or numpy."""
        if isinstance(array, np.ndarray):
            self.members[filename] = (npy_header(array.shape, False, array.dtype), array)
        else:
            self.members[filename] = (npy_header((array.numel(),), False, dtype or self._npdtype(array.element_size())), array)

    def member_bytes(self, name):
        """One member as the bytes numpy.save would have written (tests, --test sizing)."""
        header, payload = self.members[name]
        if not isinstance(payload, np.ndarray): payload = self.ctx.to_numpy(payload)
        return header + payload.tobytes()

    def compressed_size(self, data):
        """uq.py:277-285 (Q4/Q26 fixed: spawn on demand, serialise first, surface errors)."""
        if not self.args.compressor: return len(data)
        p = subprocess.run(self.args.compressor + ' | wc -c', shell=True, input=data, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        if p.returncode != 0: error('ERROR: the compressor command failed')
        self.last_subprocess_used += 1
        return int(p.stdout.split()[0])

    def test_patterns(self, table, filename):
        """uq.py:290-334: size of each candidate layout -> [size, pattern] of the smallest."""
        args = self.args
        t, rows, cols = table
        if args.compressor is None: return [rows * cols, '0.1']
        if filename.startswith('DNA'): pattern = None if args.pattern is None else args.pattern[0]
        elif filename.startswith('QUAL'): pattern = None if args.pattern is None else args.pattern[1]
        else: error('ERROR: This should never happen!')
        results = []
        for pat in (PATTERNS if pattern is None else [pattern]):
            payload = self.ops.pattern(self.ctx, t, rows, cols, pat)
            blob = pattern_header(rows, cols, pat) + self.ctx.to_numpy(payload).tobytes()
            results.append([self.compressed_size(blob), pat])
        return sorted(results)[0]

    # ------------------------------------------------------------------ table builds
    def _npdtype(self, itemsize):
        return {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[itemsize]

    def encode_dna_qual(self, sort_order, table_name, raw, test):
        """uq.py:765-805.  sort_order: None = no sort, False = compute and return, tensor = apply."""
        ops, ctx = self.ops, self.ctx
        t, rows, cols = self.tables[table_name]
        if raw:
            out_name = table_name + '.raw'
            if sort_order is None:
                table = (t, rows, cols)
            else:
                if sort_order is False: sort_order = ops.argsort_rows(ctx, t, rows, cols)       # uq.py:773-775
                table = (ops.gather_rows(ctx, t, rows, cols, sort_order), rows, cols)            # uq.py:777
            if test: test[out_name] = self.test_patterns(table, out_name)
            else: self.write_pattern(table, out_name)
        else:
            out_name = table_name + '.key'
            perm, key, skey, uniq, nu = ops.unique_rows(ctx, t, rows, cols, want_key=sort_order is not False,
                                                        want_sorted_key=sort_order is False)    # uq.py:784-789
            isz = ops.key_itemsize(nu - 1)                                                       # uq.py:790
            if sort_order is None:
                k = ops.narrow(ctx, key, isz)
            elif sort_order is False:
                sort_order = perm                                                                # == argsort(key), stable (uq.py:796)
                k = ops.narrow(ctx, skey, isz)
            else:
                k = ops.narrow(ctx, ops.gather_rows(ctx, key.view(ctx.torch.uint8), rows, 4, sort_order).view(ctx.torch.int32), isz)
            if test:
                k_host = ctx.to_numpy(k, self._npdtype(isz))
                test[out_name] = self.compressed_size(npy_header(k_host.shape, False, k_host.dtype) + k_host.tobytes())
            else: self.write_out(k, out_name, self._npdtype(isz))
            table = (uniq, nu, cols)
            if test: test[table_name] = self.test_patterns(table, table_name)
            else: self.write_pattern(table, table_name)
        return sort_order

    def encode_qname(self, sort_order, raw, test):
        """uq.py:808-851 on the device: columns stacked into big-endian rows, then the row kernels."""
        ops, ctx, columns = self.ops, self.ctx, self.columns
        cols_d = self.tables['QNAME']
        n = self.total
        common = max(c.element_size() for c in cols_d)
        ncols = len(cols_d)

        def emit(name, tensor, dtype):
            if test:
                host = ctx.to_numpy(tensor, np.dtype(dtype))
                test[name] = self.compressed_size(npy_header(host.shape, False, host.dtype) + host.tobytes())
            else: self.write_out(tensor, name, np.dtype(dtype))

        if raw:
            if sort_order is False:
                rows = ops.stack_columns(ctx, cols_d, common)                                    # uq.py:814-815
                sort_order = ops.argsort_rows(ctx, rows, n, ncols * common)                      # uq.py:816
            for idx, column in enumerate(columns):
                c = cols_d[idx]
                if sort_order is not None:
                    c = ops.gather_rows(ctx, c.view(ctx.torch.uint8), n, c.element_size(), sort_order).view(c.dtype)
                emit(column['name'] + '.raw', c, column['dtype'])
        else:
            rows = ops.stack_columns(ctx, cols_d, common)                                        # uq.py:828-829
            perm, key, skey, uniq, nu = ops.unique_rows(ctx, rows, n, ncols * common, want_key=sort_order is not False,
                                                        want_sorted_key=sort_order is False)    # uq.py:830
            isz = ops.key_itemsize(nu - 1)                                                       # uq.py:832
            if sort_order is False:
                sort_order = perm                                                                # uq.py:833
                k = ops.narrow(ctx, skey, isz)
            elif sort_order is None:
                k = ops.narrow(ctx, key, isz)
            else:
                k = ops.narrow(ctx, ops.gather_rows(ctx, key.view(ctx.torch.uint8), n, 4, sort_order).view(ctx.torch.int32), isz)
            emit('QNAME.key', k, self._npdtype(isz))
            for idx, column in enumerate(columns):                                               # uq.py:845-847
                col = ops.unstack_column(ctx, uniq, nu, ncols, common, idx, np.dtype(column['dtype']).itemsize)
                emit(column['name'], col, column['dtype'])
        return sort_order

    def run_mix(self, sorted_on, raw_tables, test):
        """uq.py:739-762: the order of the three table builds and who produces / consumes sort_order."""
        if test: test = {'sorted_on': sorted_on, 'raw_tables': raw_tables}
        if sorted_on in ['DNA', 'QUAL']:
            not_sorted_on = 'DNA' if sorted_on == 'QUAL' else 'QUAL'
            sort_order = self.encode_dna_qual(False, sorted_on, sorted_on in raw_tables, test)
            self.encode_dna_qual(sort_order, not_sorted_on, not_sorted_on in raw_tables, test)
            self.encode_qname(sort_order, 'QNAME' in raw_tables, test)
        elif sorted_on == 'QNAME':
            sort_order = self.encode_qname(False, 'QNAME' in raw_tables, test)
            self.encode_dna_qual(sort_order, 'DNA', 'DNA' in raw_tables, test)
            self.encode_dna_qual(sort_order, 'QUAL', 'QUAL' in raw_tables, test)
        else:
            self.encode_qname(None, 'QNAME' in raw_tables, test)
            self.encode_dna_qual(None, 'DNA', 'DNA' in raw_tables, test)
            self.encode_dna_qual(None, 'QUAL', 'QUAL' in raw_tables, test)
        if test:
            total_size = 0
            for key, value in test.items():
                if key.startswith('QNAME') or key.endswith('key'): total_size += value
                elif key.startswith('DNA') or key.startswith('QUAL'): total_size += value[0]
            test['total_size'] = total_size
        return test

    def run_tests(self):
        """uq.py:855-889: the (raw set x sort) grid, best-of by total size."""
        args, say = self.args, self.say
        all_results = []
        say('Starting tests...')
        say('Time:                       Sort:      Raw Tables:                      Patterns:')
        self.split_time()
        raw_grid = [('DNA', 'QUAL', 'QNAME'), ('DNA', 'QUAL'), ('QUAL', 'QNAME'), ('DNA', 'QNAME'), ('DNA',), ('QUAL',), ('QNAME',), (None,)]
        for raw_tables in (raw_grid if args.raw is None else [args.raw]):
            if args.compressor is None: args.sort = (None,)
            for to_sort in (['DNA', 'QUAL', 'QNAME', None] if args.sort is None else [args.sort]):
                all_results.append(self.run_mix(to_sort, raw_tables, True))
                say(self.split_time().ljust(27), str(to_sort).ljust(10), str(tuple(raw_tables)).ljust(32), 'All' if args.raw is None else str(args.pattern))
        best = sorted(all_results, key=lambda k: k['total_size'])[0]
        args.sort = best['sorted_on']
        args.raw = best['raw_tables']
        args.pattern = (best['DNA.raw'][1] if 'DNA.raw' in best else best['DNA'][1],
                        best['QUAL.raw'][1] if 'QUAL.raw' in best else best['QUAL'][1])
        say('\nAll done!')
        say('Size (compressed)    Sort:    Raw Tables:                 Raw stats:' if args.compressor else
            '             Size    Sort:    Raw Tables:                 Raw stats:')
        for result in sorted(all_results, key=lambda k: k['total_size']):
            rest = {k: v for k, v in result.items() if k not in ('total_size', 'sorted_on', 'raw_tables')}
            say(str(result['total_size']).rjust(17) + '   ', str(result['sorted_on']).ljust(8), str(tuple(result['raw_tables'])).ljust(27),
                ' '.join(str(k) + ':' + str(v) for k, v in rest.items()))
        # uq.py:885-889 (py2 `print x,` soft spaces kept: the line reads the same)
        line = 'Parameters found to be the best for this data type:\n  '
        if args.sort is not None and args.sort != (None,): line += '  --sort ' + str(args.sort)
        if args.raw is not None: line += '  --raw ' + ' '.join(map(str, args.raw))
        if args.pattern is not None: line += '  --pattern ' + ' '.join(map(str, args.pattern))
        say(line)
        return all_results

    # ------------------------------------------------------------------ container
    def write_container(self, path):
        """uq.py:897-913: config.json + members into an uncompressed tar (members in numeric order, Q6)."""
        args = self.args
        cfg = dict(self.config)
        cfg['sort'] = args.sort if isinstance(args.sort, str) else [None]
        cfg['raw'] = sorted(args.raw, key=str) if args.raw else [None]
        cfg['pattern'] = list(args.pattern) if args.pattern else None
        self.config = cfg
        blob = json.dumps(cfg, indent=4, sort_keys=True).encode()

        def order(name):
            base = name.split('.')[0]
            if base.startswith('QNAME_'): return (3, int(base[6:]), name)
            return ({'DNA': 0, 'QUAL': 1, 'QNAME': 2}.get(base, 4), 0, name)

        # private temp dir (Q5); beside the output unless --temp says otherwise, so the final move is a rename
        tmpdir = tempfile.mkdtemp(prefix='uq_', dir=args.temp if getattr(args, 'temp', None) else (os.path.dirname(os.path.abspath(path)) or None))

        def pwrite_all(fd, data, pos):
            mv = memoryview(data).cast('B')
            done = 0
            while done < len(mv): done += os.pwrite(fd, mv[done:], pos + done)

        try:
            tmp = os.path.join(tmpdir, 'temp.uq')
            # The tar stream tarfile.open(mode='w').addfile() would write (header block, data, padding to 512,
            # two zero blocks, padding to RECORDSIZE), with the payloads streamed HBM -> pinned -> pwrite.
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                pos = 0
                for name, (header, payload) in [('config.json', (blob, np.zeros(0, np.uint8)))] + [(k, self.members[k]) for k in sorted(self.members, key=order)]:
                    nbytes = payload.nbytes if isinstance(payload, np.ndarray) else payload.numel() * payload.element_size()
                    ti = tarfile.TarInfo(name); ti.size = len(header) + nbytes; ti.mtime = int(time.time())
                    head = ti.tobuf(tarfile.DEFAULT_FORMAT, tarfile.ENCODING, 'surrogateescape') + header
                    pwrite_all(fd, head, pos); pos += len(head)
                    if isinstance(payload, np.ndarray):
                        if nbytes: pwrite_all(fd, np.ascontiguousarray(payload), pos)
                    else:
                        self.io.device_to_fd(payload, fd, pos)
                    pos += nbytes
                    pad = -ti.size % tarfile.BLOCKSIZE
                    if pad: pwrite_all(fd, b'\0' * pad, pos); pos += pad
                end = 2 * tarfile.BLOCKSIZE
                end += -(pos + end) % tarfile.RECORDSIZE
                pwrite_all(fd, b'\0' * end, pos)
            finally:
                os.close(fd)
            shutil.move(tmp, path)
        finally:
            shutil.rmtree(tmpdir, ignore_errors=True)

    # ------------------------------------------------------------------ whole encode
    def encode(self):
        args = self.args
        if args.output is None: args.output = args.input + '.uQ'
        self.say('Warming up...')
        self.load(args.input)
        self.encode_loaded()

    def encode_loaded(self, write=True):
        """Everything after the FASTQ is in HBM (`load` / `load_device`).  write=False: stop before the container is
        written -- the members stay in `self.members` (header bytes, device payload)."""
        args = self.args
        self.analyse()
        if args.peek:
            self.say('The config.json would look like:')
            self.say(json.dumps(self.config, indent=4, sort_keys=True))
            return
        self.pack()
        if args.test: self.run_tests()
        if args.sort is None: args.sort = (None,)                                               # uq.py:893-895
        if args.raw is None: args.raw = (None,)
        self.members = {}
        self.run_mix(args.sort, args.raw, False)
        if not write: return
        self.say('\nWriting final config...')
        self.say('Archiving results and cleaning up temp directory...')
        self.write_container(args.output)
        self.say('All Done! :) ')

    # ------------------------------------------------------------------ decode (uq.py:926-1058)
    def load_from_tar(self, members, file_name, pattern='0.1', rows=None):
        """uq.py:943-945 on the device: payload -> table.  `members`: name -> (offset, size) inside the tar
        `self.tar_path`; the payload streams file -> pinned -> HBM.  Returns (device tensor, rows, cols) for
        2-D members; for 1-D ones a device tensor typed by width (its bytes are the member's dtype).
        `rows` = (lo, hi): only that range of rows / elements (the sharded decoder; a row-major payload is read
        as a slice of the file, any other layout whole)."""
        offset, size = members[file_name]
        with open(self.tar_path, 'rb') as fh:
            fh.seek(offset)
            f = io.BytesIO(fh.read(min(size, 65536)))
        version = np.lib.format.read_magic(f)
        shape, fortran, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        hdr = f.tell()
        # a .uQ file is untrusted input: the payload must be exactly what the header promises (numpy.load raises on a
        # short member, uq.py:944-945) before any of it is addressed on the device
        want = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize if len(shape) else np.dtype(dtype).itemsize
        if len(shape) not in (1, 2) or hdr + want != size:
            error('ERROR: member %s of this uQ file is damaged (header says %s x %s = %d bytes, the member holds %d)'
                  % (file_name, tuple(shape), np.dtype(dtype).name, want, size - hdr))
        if len(shape) == 1:
            tt = self.ctx.torch
            isz = np.dtype(dtype).itemsize
            lo, hi = rows if rows is not None else (0, shape[0])
            d_pay = self.io.file_to_device(self.tar_path, offset + hdr + lo * isz, (hi - lo) * isz)
            return d_pay.view({1: tt.uint8, 2: tt.int16, 4: tt.int32, 8: tt.int64}[isz])
        k = int(pattern[0])
        nrows, cols = (shape if k % 2 == 0 else shape[::-1])
        if rows is not None and pattern == '0.1':
            lo, hi = rows
            return (self.io.file_to_device(self.tar_path, offset + hdr + lo * cols, (hi - lo) * cols), hi - lo, cols)
        d_pay = self.io.file_to_device(self.tar_path, offset + hdr, size - hdr)
        # numpy flags 1-wide arrays as C order whatever the pattern asked for: the byte stream is the same
        t = d_pay if pattern == '0.1' else self.ops.unpattern(self.ctx, d_pay, nrows, cols, pattern)   # 0.1 is the table itself
        if rows is not None:
            lo, hi = rows
            return (t[lo * cols:hi * cols].contiguous(), hi - lo, cols)
        return (t, nrows, cols)

    def split_bits(self, dna, qual, config):
        """uq.py:1002-1007 + 1031-1054 on the device: rows -> characters (fixed pitch) + lengths."""
        p = self.ops.make_unpack_params(config)
        n = dna[1]
        seq, qt, ln, bad = self.ops.unpack(self.ctx, dna[0], qual[0], n, p)
        b = self.ops.bad_index(bad)
        if b is not None: error('ERROR: row %d of the DNA table carries no length sentinel; is this a uQ file?' % b)
        return seq, qt, ln

    def open_container(self):
        """Tar members (name -> (payload offset, size)) and config.json of args.input."""
        args = self.args
        if not tarfile.is_tarfile(args.input):
            error('ERROR: Sorry, the path you have provided as input is a file, but not a tar file, and therefore cannot be a .uq file!')
        self.tar_path = args.input
        with tarfile.open(args.input) as t:
            members = {m.name: (m.offset_data, m.size) for m in t.getmembers()}
            if 'config.json' not in members: error('ERROR: No config.json file was found in your input path! I cannot decode data without it!')
            config = json.loads(t.extractfile('config.json').read().decode())
        return members, config

    def load_tables(self, members, config, rows=None):
        """uq.py:943-973 on the device: the DNA / QUAL tables as (tensor, reads, row bytes) and the QNAME columns, in
        stored read order; `rows` = (lo, hi) restricts all of them to that range of reads (keys are sliced, the
        tables they index are loaded whole)."""
        ops, ctx = self.ops, self.ctx
        pat = config['pattern'] or ['0.1', '0.1']

        def table(name, pattern):
            if name + '.raw' in members: return self.load_from_tar(members, name + '.raw', pattern, rows)
            if name in members and name + '.key' in members:
                t, nrows, cols = self.load_from_tar(members, name, pattern)
                d_key = self.load_from_tar(members, name + '.key', rows=rows)
                self.check_index(d_key, nrows, name + '.key')
                return (ops.gather_rows(ctx, t, nrows, cols, d_key), d_key.numel(), cols)          # uq.py:953, 957
            error('ERROR: No ' + name + ' data was found in this uQ file?!')

        DNA = table('DNA', pat[0])
        QUAL = table('QUAL', pat[1])
        ncols = len(config['QNAME_columns'])
        u8 = ctx.torch.uint8
        if 'QNAME.key' in members:
            d_key = self.load_from_tar(members, 'QNAME.key', rows=rows)
            d_cols = []
            for i in range(ncols):                                                                # uq.py:973, numeric order (Q6)
                c = self.load_from_tar(members, 'QNAME_%d' % (i + 1))
                if i == 0: self.check_index(d_key, c.numel(), 'QNAME.key')
                d_cols.append(ops.gather_rows(ctx, c.view(u8), c.numel(), c.element_size(), d_key).view(c.dtype))
        else:
            d_cols = [self.load_from_tar(members, 'QNAME_%d.raw' % (i + 1), rows=rows) for i in range(ncols)]
        for i, c in enumerate(config['QNAME_columns']):                                           # uq.py:1016 `column['map'][row[i]]`
            if c['format'] == 'mapping': self.check_index(d_cols[i], len(c['map']), 'QNAME column %d' % (i + 1))
        if len({DNA[1], QUAL[1]} | {c.numel() for c in d_cols}) != 1:
            error('ERROR: the tables of this uQ file do not hold the same number of reads')
        return DNA, QUAL, d_cols

    def check_index(self, d_index, limit, what):
        """A stored key / mapping code must address its table: numpy raises IndexError in the reference (uq.py:953-973,
        1016); here the device check runs before the gather / the text kernels use the value."""
        bad = self.ops.check_index_range(self.ctx, d_index, limit)
        if bad is not None:
            error('ERROR: %s of this uQ file is damaged: entry %d points beyond its table of %d rows' % (what, bad, limit))

    @staticmethod
    def device_text_possible(config):
        return len(config['QNAME_columns']) <= 32 and len(config['QNAME_prefix']) <= 256 and len(config['QNAME_suffix']) <= 256

    def decode_text(self, config, DNA, QUAL, d_cols):
        """uq.py:1002-1058 on the device: the FASTQ text of these reads as one uint8 device tensor."""
        ops, ctx, n = self.ops, self.ctx, DNA[1]
        if getattr(self.args, 'two_pass_decode', False):
            seq, qt, ln = self.split_bits(DNA, QUAL, config)
            return ops.emit_fastq(ctx, config, d_cols, seq, qt, ln, n)
        # rows -> text in one kernel (uq_decode_fastq)
        text, bad = ops.decode_fastq(ctx, config, d_cols, DNA[0], QUAL[0], n)
        if bad is not None: error('ERROR: row %d of the DNA table carries no length sentinel; is this a uQ file?' % bad)
        return text

    def decode(self, out=None):
        from . import qname
        ctx = self.ctx
        out = out or sys.stdout
        members, config = self.open_container()
        DNA, QUAL, d_cols = self.load_tables(members, config)
        n = DNA[1]
        w = out.buffer if hasattr(out, 'buffer') else out
        if self.device_text_possible(config):
            # the text streams out through the pinned buffers
            self.io.device_to_stream(self.decode_text(config, DNA, QUAL, d_cols), w)
        else:
            seq, qt, ln = self.split_bits(DNA, QUAL, config)
            dmax = config['dna_max']
            S = ctx.to_numpy(seq).reshape(n, dmax); Q = ctx.to_numpy(qt).reshape(n, dmax); L = ctx.to_numpy(ln, np.uint32)
            cols = [ctx.to_numpy(c, np.dtype(cc['dtype'])) for c, cc in zip(d_cols, config['QNAME_columns'])]
            names = qname.decode_names(config, cols)
            for r in range(n):
                l = int(L[r])
                w.write(names[r].encode('latin-1') + b'\n' + S[r, :l].tobytes() + b'\n+\n' + Q[r, :l].tobytes() + b'\n')


# ---------------------------------------------------------------------- the packers with the reference's own signature
class DeviceLib:
    """What the reference's third return value (`lib`, the cffi handle of libc) is used for: `lib.free(ptr)`
    (uq.py:712-713).  Here `ptr` is a device address; free() drops the torch tensor that owns it."""

    def __init__(self):
        self._owned = {}

    def _own(self, tensor):
        self._owned[tensor.data_ptr()] = tensor
        return tensor.data_ptr()

    def free(self, ptr):
        self._owned.pop(int(ptr), None)
        return 0


def _encoder(variable, total_reads, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row, status, file_path,
             bits_per_base, bits_per_quality, ctx=None):
    from . import ops
    from .device import Context
    from .hostio import Staging
    ctx = ctx or Context(0)
    d_buf = Staging(ctx).file_to_device(file_path)
    nlines = ops.count_lines(ctx, d_buf)
    if nlines // 4 != total_reads: error('ERROR: %s holds %d reads, not %d' % (file_path, nlines // 4, total_reads))
    d_ls = ops.index_lines(ctx, d_buf, nlines)
    st = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st, d_buf, d_ls, 0, total_reads)                # dna_max and the tile sizing (the reference
    hs = ops.stats_fetch(ctx, st)                                             # closes over the global `dna_max`, uq.py:128)
    p = ops.make_pack_params(bases, qualities, N_qual, bits_per_base, bits_per_quality, variable, dna_bytes_per_row,
                             quality_bytes_per_row, hs.len_max, hs.max_record_bytes, avg_record_bytes=d_buf.numel() // max(total_reads, 1))
    dna, qual, bad = ops.pack(ctx, d_buf, d_ls, 0, total_reads, p)
    b = ops.bad_index(bad)
    if b is not None: error('ERROR: read %d holds a symbol that is in neither alphabet nor N_qual' % b)
    if status is not None: status.current = total_reads                      # uq.py:176
    lib = DeviceLib()
    return ((dna.view(total_reads, dna_bytes_per_row), lib._own(dna)),
            (qual.view(total_reads, quality_bytes_per_row), lib._own(qual)), lib)


def encoder_fixed(total_reads, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row, status, file_path,
                  bits_per_base, bits_per_quality, ctx=None):
    """uq.py:108-182 with its argument list and its return value `((dna_array, dna_ptr), (qual_array, qual_ptr), lib)`:
    the arrays are uint8[total_reads][bytes_per_row] DEVICE tensors (the reference's numpy views of malloc'd memory,
    uq.py:178-181), the pointers their device addresses, and `lib.free(ptr)` releases them (uq.py:712-713)."""
    return _encoder(False, total_reads, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row, status, file_path,
                    bits_per_base, bits_per_quality, ctx)


def encoder_variable(total_reads, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row, status, file_path,
                     bits_per_base, bits_per_quality, ctx=None):
    """uq.py:188-254, same contract as `encoder_fixed` (rows carry the length sentinel)."""
    return _encoder(True, total_reads, bases, qualities, N_qual, dna_bytes_per_row, quality_bytes_per_row, status, file_path,
                    bits_per_base, bits_per_quality, ctx)


def main(argv=None):
    args = build_parser().parse_args(argv)
    try:
        validate_args(args)
        s = Session(args)
        t_ready = time.perf_counter()                                # interpreter, torch and the device context are up
        if args.decode:
            s.decode()
        else:
            s.encode()
        if os.environ.get('UQ_TIMING'):
            s.ctx.sync()
            print(json.dumps({'uq_timing': 'uq', 'work_s': round(time.perf_counter() - t_ready, 3)}), file=sys.stderr, flush=True)
    except UqError as e:
        print(e)
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
