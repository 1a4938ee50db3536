#!/usr/bin/env python3
"""Round 0 of the row sort, LSD passes against the MSD partition (csrc/msd.hip) and its level plans: argsort + unique of the DNA and the QUAL table
(BASELINE configs[2] shape: a tenth of the reads copy one of N / 16 templates).
    python tools/msdbench.py [reads] [plans: e.g. lsd auto 8,5,5 8,8 9,9]  (UQ_MSDBENCH_ONE=DNA|QUAL: one table, for kernel traces)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import analysis, ops, synth
from uq_amd.device import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
plans = sys.argv[2:] or ['lsd', 'auto']
spec = synth.Spec(20261003 + 4, 150, dup='both', dup_templates=max(1, n // 16))
d_buf = ops.synth_fastq(ctx, spec, 0, n)
nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
st = ops.stats_new(ctx); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); hs = ops.stats_fetch(ctx, st)
d = analysis.decide_from_stats(hs)
p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                         d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
del d_buf, ls
torch.cuda.empty_cache()
tables = {'DNA': (dna, d['dna_bytes_per_row']), 'QUAL': (qual, d['quality_bytes_per_row'])}
one = os.environ.get('UQ_MSDBENCH_ONE')
ref = {}
for plan in plans:
    if plan == 'lsd': ops.sort_config(ctx, msd_min_rows=-1)
    elif plan == 'auto': ops.sort_config(ctx)
    else: ops.sort_config(ctx, level_bits=[int(x) for x in plan.split(',')])
    for name, (tab, C) in tables.items():
        if one and name != one: continue
        for op, fn in (('argsort', lambda: ops.argsort_rows(ctx, tab, n, C)), ('unique', lambda: ops.unique_rows(ctx, tab, n, C)[3:])):
            best, out = None, None
            c0 = ops.sort_counters(ctx)
            for _ in range(3):
                out = None
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1); best = ms if best is None or ms < best else best
            c1 = ops.sort_counters(ctx)
            same = ''
            if op == 'argsort':
                if (name, op) in ref: same = ' same order: %s' % bool(torch.equal(ref[(name, op)], out))
                else: ref[(name, op)] = out
            print('%-8s %-5s %-8s %d x %3d B: %8.3f ms   (msd %d, lsd %d of 3)%s' % (plan, name, op, n, C, best, c1[0] - c0[0], c1[1] - c0[1], same), flush=True)
            del out
