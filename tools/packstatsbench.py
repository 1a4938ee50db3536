#!/usr/bin/env python3
"""uq_pack_stats (pack + pass-1 statistics in one read, counted on the codes) against uq_stats_accumulate + uq_pack.
    python tools/packstatsbench.py [reads] [cfg5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from uq_amd import analysis, ops, synth
from uq_amd.device import Context

ctx = Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
spec = synth.Spec(20261003 + 5, (36, 301), n_rate=1) if 'cfg5' in sys.argv else synth.Spec(20261005, 150)
d_buf = ops.synth_fastq(ctx, spec, 0, n)
nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
st = ops.stats_new(ctx); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); hs = ops.stats_fetch(ctx, st)
d = analysis.decide_from_counts(hs.counts, hs.len_min, hs.len_max)
p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                         d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes, avg_record_bytes=d_buf.numel() // n)

def timed(fn, reps=6):
    best, out = None, None
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); best = ms if best is None or ms < best else best
    return best, out

def stats():
    s = ops.stats_new(ctx); ops.stats_accumulate(ctx, s, d_buf, ls, 0, n); return s
t_s, _ = timed(stats)
t_p, (dna, qual, bad) = timed(lambda: ops.pack(ctx, d_buf, ls, 0, n, p))
t_f, res = timed(lambda: ops.pack_stats(ctx, d_buf, ls, 0, n, p))
ok = res is not None and torch.equal(res[0], dna) and torch.equal(res[1], qual)
h2 = ops.stats_fetch(ctx, res[3]) if res is not None else None
same = h2 is not None and not h2.incomplete and np.array_equal(h2.counts, hs.counts) and (h2.len_min, h2.len_max, h2.max_record_bytes) == (hs.len_min, hs.len_max, hs.max_record_bytes)
print('stats %.3f ms + pack %.3f ms = %.3f ms; pack_stats %.3f ms (tables identical: %s, statistics identical: %s)' % (t_s, t_p, t_s + t_p, t_f, ok, same))

# ... and with the QNAME phase (uq_pack_stats_qname): the guess is made once, outside the timing
fq = ops.FusedQname(ctx, n)
ops.qname_guess(ctx, d_buf, ls, n, fq)
st_pre = [ops.stats_new(ctx) for _ in range(8)]
def fused():
    return ops.pack_stats(ctx, d_buf, ls, 0, n, p, fq=fq)
t_q, resq = timed(fused)
okq = resq is not None and torch.equal(resq[0], dna) and torch.equal(resq[1], qual)
print('pack_stats + QNAME phase %.3f ms (tables identical: %s)%s' % (t_q, okq, ''.join(' %s=%s' % (k, v) for k, v in os.environ.items() if k.startswith('UQ_'))))
