#!/usr/bin/env python3
"""Row sort alone (BASELINE configs[2] shape): argsort + unique of the N x 38 B DNA table, for rocprofv3 kernel traces.
    python tools/sortbench.py [reads] [DNA|QUAL]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import analysis, ops, synth
from uq_amd.device import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
which = sys.argv[2] if len(sys.argv) > 2 else 'DNA'
spec = synth.Spec(20261003 + 3, 150, dup='dna' if which == 'DNA' else 'qual', dup_templates=max(1, n // 16))
d_buf = ops.synth_fastq(ctx, spec, 0, n)
nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
st = ops.stats_new(ctx); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); hs = ops.stats_fetch(ctx, st)
d = analysis.decide_from_counts(hs.counts, hs.len_min, hs.len_max)
p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                         d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
del d_buf, ls
tab, C = (dna, d['dna_bytes_per_row']) if which == 'DNA' else (qual, d['quality_bytes_per_row'])
for name, fn in (('argsort_rows', lambda: ops.argsort_rows(ctx, tab, n, C)), ('unique_rows', lambda: ops.unique_rows(ctx, tab, n, C))):
    best = None
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None or ms < best else best
    print('%s[%s] %d x %d B: %.3f ms' % (name, which, n, C, best), flush=True)
