#!/usr/bin/env python3
"""Where the bench step's time goes, stage by stage (each stage synchronised: the sum is an upper bound of the pipelined step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import analysis, ops, synth
from uq_amd.device import Context
ctx = Context(0)
n = 10_000_000
d_buf = ops.synth_fastq(ctx, synth.Spec(20261005, 150), 0, n)
acc = {}
def lap(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
K = 20
for it in range(K + 3):
    if it == 3: acc.clear()
    nl = lap('count_lines', lambda: ops.count_lines(ctx, d_buf))
    ls = lap('index_lines', lambda: ops.index_lines(ctx, d_buf, nl))
    g = lap('head_guess_indexed', lambda: ops.head_guess_indexed(ctx, d_buf, ls, n))
    spec = lap('pack_stats', lambda: ops.pack_stats(ctx, d_buf, ls, 0, n, g))
    hs = lap('stats_fetch', lambda: ops.stats_fetch(ctx, spec[3]))
    d = lap('decide+params', lambda: analysis.decide_from_counts(hs.counts, hs.len_min, hs.len_max))
print({k: round(v / K * 1e3, 3) for k, v in acc.items()}, 'sum ms', round(sum(acc.values()) / K * 1e3, 3))
