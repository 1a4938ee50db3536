#!/usr/bin/env python3
"""Times the decode kernels on one GPU: uq_unpack and uq_emit_fastq over tables packed
from the synthetic generator (the bytes of BASELINE configs[1]: N x 150 bp), and checks that the emitted text is the
input FASTQ again (the round trip needs no oracle at this size).

    python tools/bench_decode.py [--reads 10000000] [--length 150] [--var-min 0] [--reps 5]

Prints one JSON line per operation: ms, algorithmic GB/s (bytes read + bytes written).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from uq_amd import analysis, ops, qname_device, synth
from uq_amd.device import Context


def timed(fn, reps):
    best, out = None, None
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=10_000_000)
    ap.add_argument('--length', type=int, default=150)
    ap.add_argument('--var-min', type=int, default=0, help='variable read lengths from this minimum (0 = fixed)')
    ap.add_argument('--n-rate', type=int, default=0)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--notricks', action='store_true', help='no N-trick: N stays a base (3-bit DNA), the alphabets go through the tables')
    ap.add_argument('--no-check', action='store_true', help='timing experiments with variant libraries (UQ_LIB_PATH): a wrong text is reported, not fatal')
    ap.add_argument('--only-fused', action='store_true', help='time uq_decode_fastq only (clean kernel / counter profiles)')
    args = ap.parse_args()
    ctx = Context(0)
    n = args.reads
    spec = synth.Spec(20261003 + 1, (args.var_min, args.length) if args.var_min else args.length, n_rate=args.n_rate)
    d_buf = ops.synth_fastq(ctx, spec, 0, n)
    nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
    st = ops.stats_new(ctx); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); hs = ops.stats_fetch(ctx, st)
    d = analysis.decide_from_stats(hs, notricks=args.notricks)
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
    dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
    config = dict(d)
    up = ops.make_unpack_params(config)

    def report(op, ms, nbytes, **kw):
        print(json.dumps(dict(op=op, reads=n, ms=round(ms, 3), algorithmic_GBps=round(nbytes / 1e9 / (ms / 1e3), 1), **kw)), flush=True)

    if not args.only_fused:
        ms, (seq, qt, ln, ubad) = timed(lambda: ops.unpack(ctx, dna, qual, n, up), args.reps)
        report('unpack', ms, dna.numel() + qual.numel() + seq.numel() + qt.numel() + 4 * n, dna_max=d['dna_max'],
               bits=[d['bits_per_base'], d['bits_per_quality']])
    # QNAME columns the way the encoder stores them (device analysis: layout, typing, values)
    import time
    t0 = time.perf_counter()
    prefix, suffix, separators, columns, arrays = qname_device.analyse_device(ctx, d_buf, ls, n)
    torch.cuda.synchronize()
    report('qname_analyse', (time.perf_counter() - t0) * 1e3, d_buf.numel(), columns=[c['format'] + ':' + c['dtype'] for c in columns])
    cols = [a if torch.is_tensor(a) else ctx.to_device(a) for a in arrays]
    config.update(QNAME_prefix=prefix, QNAME_suffix=suffix, QNAME_separators=separators, QNAME_columns=columns)
    if not args.only_fused:
        ms, text = timed(lambda: ops.emit_fastq(ctx, config, cols, seq, qt, ln, n), args.reps)
        same = text.numel() == d_buf.numel() and bool(torch.equal(text, d_buf))
        report('emit_fastq', ms, seq.numel() + qt.numel() + 4 * n + sum(c.numel() * c.element_size() for c in cols) + text.numel(), round_trip=same,
               text_bytes=text.numel())
        if not same: sys.exit('decode bench: the emitted text differs from the input')
        del text, seq, qt
        torch.cuda.empty_cache()
    ms, (text, bad) = timed(lambda: ops.decode_fastq(ctx, config, cols, dna, qual, n), args.reps)
    same = bad is None and text.numel() == d_buf.numel() and bool(torch.equal(text, d_buf))
    report('decode_fastq', ms, dna.numel() + qual.numel() + sum(c.numel() * c.element_size() for c in cols) + text.numel(), round_trip=same)
    if not same and not args.no_check: sys.exit('decode bench: the one-pass text differs from the input')


if __name__ == '__main__':
    main()
