#!/usr/bin/env python3
"""Times the table-build kernels (BASELINE configs[2]: N x 150bp, --sort DNA + unique/index) on one GPU.

    python tools/bench_tables.py [--reads 50000000] [--sort DNA|QUAL] [--reps 3]

Prints one JSON line per operation: ms, algorithmic GB/s (SURVEY.md 8d bytes), and a property check
(sortedness of the gathered rows, key/perm consistency) that does not need the oracle at this size.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from uq_amd import analysis, ops, synth
from uq_amd.device import Context


def timed(fn, reps):
    best = None
    out = None
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=50_000_000)
    ap.add_argument('--length', type=int, default=150)
    ap.add_argument('--sort', default='DNA')
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    ctx = Context(0)
    n = args.reads
    spec = synth.Spec(20261003 + 3, args.length, dup='dna' if args.sort == 'DNA' else 'qual', dup_templates=max(1, n // 16))
    d_buf = ops.synth_fastq(ctx, spec, 0, n)
    nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
    st = ops.stats_new(ctx); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); hs = ops.stats_fetch(ctx, st)
    d = analysis.decide_from_counts(hs.counts, hs.len_min, hs.len_max)
    p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'],
                             d['dna_bytes_per_row'], d['quality_bytes_per_row'], d['dna_max'], hs.max_record_bytes)
    dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, n, p)
    del d_buf, ls
    torch.cuda.empty_cache()
    Cd, Cq = d['dna_bytes_per_row'], d['quality_bytes_per_row']
    tab, C, other, Co = (dna, Cd, qual, Cq) if args.sort == 'DNA' else (qual, Cq, dna, Cd)

    def report(op, ms, nbytes, **kw):
        print(json.dumps(dict(op=op, reads=n, ms=round(ms, 3), algorithmic_GBps=round(nbytes / 1e9 / (ms / 1e3), 1), **kw)), flush=True)

    ms, perm = timed(lambda: ops.argsort_rows(ctx, tab, n, C), args.reps)
    report('argsort_rows[%s]' % args.sort, ms, n * (C + 4), cols=C)
    ms, g = timed(lambda: ops.gather_rows(ctx, tab, n, C, perm), args.reps)
    report('gather_rows[%s]' % args.sort, ms, n * (2 * C + 4), cols=C)
    # property: gathered rows are sorted (compare neighbours on the first 8 bytes + full rows on a sample)
    G = g.view(n, C)
    k = min(n - 1, 2_000_000)
    a = G[:k].cpu().numpy(); b = G[1:k + 1].cpu().numpy()
    neq = a != b
    first = neq.argmax(axis=1)
    ok = (~neq.any(axis=1)) | (a[np.arange(k), first] < b[np.arange(k), first])
    report('check_sorted_prefix', 0.001, 0, ok=bool(ok.all()))
    del g, G
    ms, go = timed(lambda: ops.gather_rows(ctx, other, n, Co, perm), args.reps)
    report('gather_rows[other]', ms, n * (2 * Co + 4), cols=Co)
    del go
    ms, (perm2, key, skey, uniq, nu) = timed(lambda: ops.unique_rows(ctx, tab, n, C), args.reps)
    report('unique_rows[%s]' % args.sort, ms, n * (C + 4) + nu * C, cols=C, nunique=nu)
    assert torch.equal(perm2, perm), 'unique_rows order != argsort_rows order'
    # property: key is consistent with the order (key[perm] non-decreasing, ends at nu-1)
    sk = skey.cpu().numpy().view(np.uint32)
    assert sk[0] == 0 and sk[-1] == nu - 1 and (np.diff(sk.astype(np.int64)) >= 0).all() and (np.diff(sk.astype(np.int64)) <= 1).all()
    isz = ops.key_itemsize(nu - 1)
    ms, _ = timed(lambda: ops.narrow(ctx, skey, isz), args.reps)
    report('narrow_key', ms, n * (4 + isz), itemsize=isz)
    for pat in ('0.2', '1.2', '2.2', '3.1'):
        ms, pay = timed(lambda: ops.pattern(ctx, other, n, Co, pat), args.reps)
        report('pattern[%s] %dx%d' % (pat, n, Co), ms, 2 * n * Co)
        ms, back = timed(lambda: ops.unpattern(ctx, pay, n, Co, pat), args.reps)
        report('unpattern[%s]' % pat, ms, 2 * n * Co, roundtrip=bool(torch.equal(back, other)))
        del pay, back
    cfg = dict(bases=d['bases'], qualities=d['qualities'], N_qual=d['N_qual'], bits_per_base=d['bits_per_base'], bits_per_quality=d['bits_per_quality'],
               variable_read_lengths=d['variable_read_lengths'], dna_max=d['dna_max'])
    up = ops.make_unpack_params(cfg)
    ms, _ = timed(lambda: ops.unpack(ctx, dna, qual, n, up), args.reps)
    report('unpack', ms, n * (Cd + Cq + 2 * d['dna_max'] + 4))


if __name__ == '__main__':
    main()
