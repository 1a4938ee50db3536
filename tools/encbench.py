#!/usr/bin/env python3
"""Times uq_encode_stream alone on BASELINE configs[1] (10 M x 150 bp in HBM): with / without the statistics.
    python tools/encbench.py [reads]        (UQ_ENC_DEBUG=1: no look-back -- timing experiment, results are wrong)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import ops, synth
from uq_amd.device import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d_buf = ops.synth_fastq(ctx, synth.Spec(20261005, 150), 0, n)
guess, rpb = ops.head_guess(ctx, d_buf)
for with_stats in (True, False):
    best = None
    for _ in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); e = ops.encode_stream(ctx, d_buf, guess, int(n * 1.01) + 8, with_stats=with_stats); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    print('encode_stream stats=%s: %.3f ms  (nlines %d, index %s, tables %s)' % (with_stats, best, e.nlines, e.line_start is not None, e.tables is not None), flush=True)
