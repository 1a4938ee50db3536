"""One rank of the sharded fused QNAME pass on a real file (run under torch.distributed.run by tests/test_gpu_dist.py; ranks share the card over
gloo): the file's records are dealt to the ranks in contiguous ranges, rank 0 guesses the layout, ONE broadcast, every rank's pack kernel verifies
and parses its reads, qname_device.analyse_fused_sharded combines -- and every rank's columns, concatenated, must be the oracle's
(oracle/uq_oracle.py on the whole file), or the pass must have stood down (printed as DECLINED; the caller says which it expects)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle')); sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.distributed as dist

import oracle_c
import uq_oracle as O
from uq_amd import dist as uqdist, ops, qname_device
from uq_amd.device import Context


def main():
    path = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ctx = Context(0)
    host = np.fromfile(path, dtype=np.uint8)
    hls = oracle_c.index_lines(host)
    N = (len(hls) - 1) // 4
    lo, hi = uqdist.shard_range(N, rank, world)
    b0, b1 = int(hls[4 * lo]), int(hls[4 * hi])
    n = hi - lo
    shard = uqdist.Shard(uqdist.HipRows(ctx), lo, N)
    buf = ctx.to_device(host[b0:b1].copy())
    res, ls, guess = None, None, None
    fq = ops.FusedQname(ctx, n + 8)
    if n:
        nl = ops.count_lines(ctx, buf)
        ls = ops.index_lines(ctx, buf, nl)
        guess = ops.head_guess_indexed(ctx, buf, ls, n)
    if rank == 0: ops.qname_guess(ctx, buf, ls, n, fq)
    qname_device.broadcast_guess(ctx, fq, shard)
    packed = ops.pack_stats(ctx, buf, ls, 0, n, guess, fq=fq) if n and guess is not None else None
    if packed is not None: ops.qname_fused_finish(ctx, fq)
    # (a rank whose pack kernel did not run reports it through its flags: nreads != n)
    res = qname_device.analyse_fused_sharded(ctx, fq, n, shard, usable=packed is not None or n == 0)
    out = None
    if res is not None:
        pre, suf, sep, cols, arrs = res
        out = dict(layout=[pre, suf, sep], cols=cols, arrays=[ctx.to_numpy(a, np.dtype(c['dtype'])).tolist() for a, c in zip(arrs, cols)])
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        lines = O.read_lines(host.tobytes())
        try:
            p1 = O.pass1(lines)
            ocols = O.qname_columns(lines, p1['prefix'], p1['suffix'], p1['separators'])
            oarr = O.qname_encode(lines, p1['prefix'], p1['suffix'], p1['separators'], ocols)
            want = dict(layout=[p1['prefix'], p1['suffix'], p1['separators']], cols=ocols)
        except Exception as e:
            want, oarr = None, None
        if any(g is None for g in gathered):
            assert all(g is None for g in gathered), 'some ranks answered, some declined'
            print('DECLINED')
        else:
            assert want is not None, 'the sharded fused pass answered where the reference refuses'
            for g in gathered:
                assert g['layout'] == want['layout'] and json.loads(json.dumps(g['cols'])) == json.loads(json.dumps(want['cols'])), (g['layout'], g['cols'], want)
            for c in range(len(ocols)):
                got = np.concatenate([np.asarray(g['arrays'][c], dtype=oarr[c].dtype) for g in gathered])
                assert np.array_equal(got, oarr[c]), 'column %d differs from the oracle' % c
            print('OK %d columns' % len(ocols))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
