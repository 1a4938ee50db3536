#!/usr/bin/env python3
"""Stage by stage: the two global --sort legs of bench.py on one GPU (BASELINE configs[2] / [3] shape).
    python tools/sortlegbench.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import analysis, dist as uqdist, ops, synth
from uq_amd.device import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
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
Cd, Cq = d['dna_bytes_per_row'], d['quality_bytes_per_row']
be = uqdist.HipRows(ctx)
t = torch


def timed(name, fn, reps=3):
    best, out = None, None
    for _ in range(reps):
        out = None
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None or ms < best else best
    print('%-44s %9.3f ms' % (name, best), flush=True)
    return out


print('%d reads, DNA %d B, QUAL %d B' % (n, Cd, Cq))
pd = timed('argsort DNA', lambda: ops.argsort_rows(ctx, dna, n, Cd))
pq = timed('argsort QUAL', lambda: ops.argsort_rows(ctx, qual, n, Cq))
gd = timed('gather DNA rows by order', lambda: ops.gather_rows(ctx, dna, n, Cd, pq))
gq = timed('gather QUAL rows by order', lambda: ops.gather_rows(ctx, qual, n, Cq, pq))
idx = timed('order -> file-wide int64 (affine)', lambda: be.index_affine(pq, 0, 8))
timed('unique of the sorted QUAL rows', lambda: ops.unique_sorted_rows(ctx, gq, n, Cq))
sd = ops.gather_rows(ctx, dna, n, Cd, pd)
timed('unique of the sorted DNA rows', lambda: ops.unique_sorted_rows(ctx, sd, n, Cd))
ids = t.arange(n, dtype=t.int32, device=ctx.device)
inv = timed('invert permutation', lambda: be.invert_permutation(idx, 0))
timed('gather 4-byte ids by the inverse', lambda: ops.gather_rows(ctx, ids.view(t.uint8), n, 4, inv))
timed('dist_scatter_rows (4-byte ids)', lambda: uqdist.dist_scatter_rows(be, ids.view(t.uint8), 4, [0, n], idx))
timed('dist_gather_rows (4-byte ids)', lambda: uqdist.dist_gather_rows(be, ids.view(t.uint8), n, 4, [0, n], idx))
timed('global_sort_rows QUAL (argsort + gather + affine)', lambda: uqdist.global_sort_rows(be, qual, n, Cq, 0, total_rows=n))
timed('global_sort_rows DNA', lambda: uqdist.global_sort_rows(be, dna, n, Cd, 0, total_rows=n))
