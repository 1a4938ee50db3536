#!/usr/bin/env python3
"""Times uq_stats_accumulate alone on the BASELINE configs[1] shape (10 M x 150 bp, generated on the device).
    python tools/statbench.py [reads]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import ops, synth
from uq_amd.device import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d_buf = ops.synth_fastq(ctx, synth.Spec(20261005, 150), 0, n)
nl = ops.count_lines(ctx, d_buf)
ls = ops.index_lines(ctx, d_buf, nl)
best = None
for _ in range(5):
    st = ops.stats_new(ctx)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    best = ms if best is None or ms < best else best
print('stats_kernel: %.3f ms for %d reads (%.0f GB/s of FASTQ)' % (best, n, d_buf.numel() / 1e6 / best))

