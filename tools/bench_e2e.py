#!/usr/bin/env python3
"""End-to-end CLI timing: page-cached FASTQ file -> .uQ tar on tmpfs (and back), SURVEY.md 8d "end-to-end".

    python tools/bench_e2e.py [--reads 10000000] [--flags "--sort None --raw DNA QUAL QNAME"] [--decode]

Writes a synth-v1 FASTQ to /dev/shm (generated on the GPU), runs `uq_amd.uq` on it in-process with a
stage timer, prints one JSON line.  PCIe transfers, the host QNAME passes, numpy/tar writing are all
inside these numbers (they are NOT bench.py's `value`, which is HBM-resident).
"""
import argparse
import hashlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from uq_amd import ops, synth, uq
from uq_amd.device import Context


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=10_000_000)
    ap.add_argument('--length', type=int, default=150)
    ap.add_argument('--flags', default='--sort None --raw DNA QUAL QNAME --pattern 0.1 0.1')
    ap.add_argument('--decode', action='store_true')
    ap.add_argument('--dir', default='/dev/shm')
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 5],
                    help='BASELINE configs[n-1] shape: 2 = fixed length; 3 = 10 %% of reads copy one of N/16 DNA templates; 5 = 36-301 bp, 1 %% N')
    args = ap.parse_args()
    ctx = Context(0)
    if args.config == 3: spec = synth.Spec(20261003 + 3, args.length, dup='dna', dup_templates=max(1, args.reads // 16))
    elif args.config == 5: spec = synth.Spec(20261003 + 5, (36, 301), n_rate=1)
    else: spec = synth.Spec(20261003 + 2, args.length)
    d = ops.synth_fastq(ctx, spec, 0, args.reads)
    path = os.path.join(args.dir, 'uq_e2e_%d.fastq' % os.getpid())
    out = path + '.uQ'
    host = ctx.to_numpy(d)
    host.tofile(path)
    nbytes = host.size
    sha = hashlib.sha256(host.tobytes()).hexdigest() if args.decode else None
    del d, host
    try:
        a = uq.validate_args(uq.build_parser().parse_args(['-i', path, '-o', out, '--quiet'] + args.flags.split()))
        s = uq.Session(a, ctx=ctx)
        stages = {}
        t0 = time.perf_counter()

        def lap(name, fn):
            t = time.perf_counter(); fn(); ctx.sync(); stages[name] = round(time.perf_counter() - t, 3)

        lap('read_file+h2d+census+pack_stats_qname(queued)', lambda: s.load(path))
        lap('stats+decisions+qname', s.analyse)
        lap('pack', s.pack)
        if a.sort is None: a.sort = (None,)
        if a.raw is None: a.raw = (None,)
        lap('tables+patterns+d2h', lambda: s.run_mix(a.sort, a.raw, False))
        lap('tar', lambda: s.write_container(out))
        enc = time.perf_counter() - t0
        res = {'op': 'cli_encode', 'reads': args.reads, 'fastq_MB': round(nbytes / 1e6, 1), 'flags': args.flags, 'seconds': round(enc, 3),
               'MBps': round(nbytes / 1e6 / enc, 1), 'reads_per_s': round(args.reads / enc), 'stages_s': stages,
               'uq_MB': round(os.path.getsize(out) / 1e6, 1), 'qname_path': s.qname_path}
        print(json.dumps(res), flush=True)
        if args.decode:
            a2 = uq.validate_args(uq.build_parser().parse_args(['-i', out, '--decode', '--quiet']))
            buf = io.BytesIO()
            t0 = time.perf_counter()
            uq.Session(a2, ctx=ctx).decode(out=buf)
            dec = time.perf_counter() - t0
            ok = hashlib.sha256(buf.getvalue()).hexdigest() == sha if a.sort == (None,) else None
            print(json.dumps({'op': 'cli_decode', 'seconds': round(dec, 3), 'MBps': round(nbytes / 1e6 / dec, 1), 'roundtrip_identical': ok}), flush=True)
    finally:
        for p in (path, out):
            if os.path.exists(p): os.remove(p)


if __name__ == '__main__':
    main()
