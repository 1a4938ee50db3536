#!/usr/bin/env python3
"""The bench step's pack + statistics (+ QNAME) call alone, queued form without a record index, no result checks: for ablation builds of
pack.hip (UQ_LIB_PATH) whose counts or tables are deliberately wrong.    python tools/packkernel_time.py [reads] [noqn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uq_amd import ops, synth
from uq_amd.device import Context

ctx = Context(0)
from uq_amd.device import SideContext
side = SideContext(ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10_000_000
qn = 'noqn' not in sys.argv
d_buf = ops.synth_fastq(ctx, synth.Spec(20261005, 150), 0, n)
nbytes = d_buf.numel()
ms = []
for it in range(12):
    st = ops.stats_new(ctx)
    census = ops.ChunkedCensus(ctx, d_buf); census.chunk(0, nbytes); census.end_async()
    guess, rpb = ops.head_guess(side, d_buf, notricks=False, head_bytes=ops.HEAD_BYTES_SMALL, head_reads=ops.HEAD_READS_INDEXED)
    cap = int(nbytes * rpb * 1.02) + 1024
    guess.avg_record_bytes = int(1.0 / rpb)
    fq = None
    if qn:
        fq = ops.FusedQname(ctx, cap); ops.qname_guess_async(ctx, d_buf, None, fq)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); sp = ops.pack_stats_async(ctx, d_buf, None, cap, guess, st=st, fq=fq); e1.record()
    census.wait(); torch.cuda.synchronize()
    if it >= 2: ms.append(e0.elapsed_time(e1))
    del sp, fq, st
print('%s: pack + statistics%s call %.4f ms (min %.4f) over %d runs' % (os.path.basename(os.environ.get('UQ_LIB_PATH', 'libuqhip.so')), ' + QNAME' if qn else '', sum(ms) / len(ms), min(ms), len(ms)))
