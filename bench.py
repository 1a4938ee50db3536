#!/usr/bin/env python3
"""bench.py -- uQ encode hot path on MI355X: FASTQ MB/s on BASELINE.json configs[1].

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): 10 M x 150 bp synthetic FASTQ ("synth-v1", seed 20261005) PER GPU, resident
in HBM before the timed region, `--sort None --raw DNA QUAL QNAME --pattern 0.1 0.1`: weak scaling,
reads are sharded record-parallel, rank r owns reads [r*10M, (r+1)*10M).
One step = one pass of the hot path over the shard, everything from the raw bytes to the packed tables:
    newline census -> record index                                                     (first read of the stream)
    -> decisions guessed from the shard's first 8192 reads (a small statistics pass of their own)
    -> uq_pack_stats: DNA 2-bit + QUAL 6-bit pack with the guess AND the pass-1 statistics (256x256 histogram, lengths,
       checks) of every read, counted on the codes the conversion produces            (second read of the stream)
    -> [N > 1: all-reduce of the statistics over RCCL, the path's only exchange without --sort]
    -> alphabet / N-trick / bit-width decisions on the host from the WHOLE shard's statistics (uq.py:448-545); the tables are
       kept iff they equal the guess, else uq_pack runs with the real ones (never on this workload).
    (--multi-pass: round 1's step, statistics and pack as separate kernels = three reads.)
    -> the QNAME passes (uq.py:394-444 layout, 555-678 column typing, 717-736 column encoding) INSIDE that
       second read: the pack kernel holds the QNAME lines in its LDS tiles anyway; a layout guessed on the device from a sample of
       the reads is verified on every read while the fields are parsed; distinct-value counts and the column encoders are queued
       behind it.  (N > 1: rank 0's layout guess is broadcast, every rank verifies it on its shard inside its pack kernel, flags / ranges / first
       occurrences are combined over the ranks in one all-gather and one all-reduce: qname.in_step is true at every N.)
`value` = FASTQ bytes of all ranks / time, MAX over ranks.
Besides the contract fields the JSON line carries `roofline` (pack kernel, HIP-event timed inside the
timed region on the launch stream) and `cpu_baseline` (the faithful per-base Python loops of the
oracle on one host core, on a bounded sample of the same input; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

SEED = 20261003 + 2
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def host_decide(hs, notricks=False, pad=False):
    from uq_amd import analysis
    return analysis.decide_from_stats(hs, notricks=notricks, pad=pad)


def cpu_baseline(sample_bytes, n_sample, d):
    """Oracle legs on the host: the per-base Python loops the reference runs (one core), and the C port."""
    sys.path.insert(0, os.path.join(HERE, 'oracle'))
    sys.path.insert(0, os.path.join(HERE, 'tests'))
    import uq_oracle as O
    t0 = time.perf_counter()
    lines = O.read_lines(sample_bytes)
    # pass-1 histogram + lengths (uq.py:369-375, 415-425), the per-base dict loop
    sq = {}
    dmin = dmax = len(lines[1]) - 1
    for r in range(n_sample):
        dna = lines[4 * r + 1][:-1]; q = lines[4 * r + 3][:-1]
        if len(dna) > dmax: dmax = len(dna)
        if len(dna) < dmin: dmin = len(dna)
        for b, c in zip(dna, q):
            try: sq[b][c] += 1
            except KeyError:
                sq.setdefault(b, {}); sq[b][c] = sq[b].get(c, 0) + 1
    # decisions: the whole-file ones of the GPU run (a sample may see fewer symbols / shorter reads)
    dna, qual = O.encoder(lines, d['bases'], d['qualities'], d['N_qual'], d['dna_bytes_per_row'], d['quality_bytes_per_row'],
                          d['bits_per_base'], d['bits_per_quality'], d['variable_read_lengths'])
    t_py = time.perf_counter() - t0
    out = {'value': round(len(sample_bytes) / 1e6 / t_py, 4), 'unit': 'MB/s', 'cores': 1, 'kind': 'port',
           'sample': 'first %d reads (%.1f MB) of the same synthetic FASTQ: line split + per-base histogram + '
                     'per-base pack loops of oracle/uq_oracle.py (CPython, one core), %.1f s' % (n_sample, len(sample_bytes) / 1e6, t_py),
           'reads_per_s': round(n_sample / t_py, 1), 'host_cores': os.cpu_count()}
    try:
        import oracle_c
        buf = np.frombuffer(sample_bytes, dtype=np.uint8)
        t0 = time.perf_counter()
        ls = oracle_c.index_lines(buf)
        st = oracle_c.stats(buf, ls, 0, n_sample)
        cd, cq, _ = oracle_c.pack(buf, ls, 0, n_sample, d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'],
                                  d['bits_per_quality'], d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'])
        t_c = time.perf_counter() - t0
        out['c_port_MBps'] = round(len(sample_bytes) / 1e6 / t_c, 2)
        out['c_port_matches_python'] = bool(np.array_equal(cd, dna) and np.array_equal(cq, qual))
    except Exception as e:  # the C leg is optional
        out['c_port_error'] = repr(e)
    return out, (dna, qual)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--reads', type=int, default=10_000_000, help='reads per GPU (default: BASELINE configs[1])')
    ap.add_argument('--length', type=int, default=150)
    ap.add_argument('--cpu-sample', type=int, default=250_000, help='reads timed on the host for cpu_baseline (0 = skip; the default is ~12 s of the Python port on one core)')
    ap.add_argument('--north-star-reads', type=int, default=200_000_000,
                    help='N = 1: the same step timed once more at the north_star\'s single-GPU size (200 M x 150 bp), reported as '
                         'north_star_200M; 0 = skip')
    ap.add_argument('--sort-reads', type=int, default=200_000_000,
                    help='reads of the global --sort legs over ALL GPUs (strong scaling: 200 M / N per GPU; BASELINE configs[3] is 200 M over '
                         '8 GPUs); 0 = skip')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end record (N = 1: the CLI on a tmpfs file of the same workload, both ways)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group and run the collectives even with one rank')
    ap.add_argument('--multi-pass', action='store_true',
                    help='round 1\'s step: census -> index -> statistics -> decisions -> pack, three reads of the stream (the default '
                         'counts the statistics inside the pack kernel: two reads)')
    ap.add_argument('--no-qname', action='store_true', help='leave the QNAME passes out of the timed step (round 2\'s step)')
    ap.add_argument('--workload', default='cfg2', choices=['cfg2', 'cfg5-notricks', 'cfg5-ntrick'],
                    help='cfg2 = BASELINE configs[1] (the bench line the driver reads); cfg5-* = configs[4]: variable length 36-301 bp '
                         'with 1%% N, 3-bit ACGNT path (--notricks) or 2-bit N-trick path -- parity/measurement extras')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))

    import torch
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist          # --force-dist: exercise the RCCL path with a single rank
    from uq_amd import ops, synth
    from uq_amd.device import Context

    json_out = sys.stdout
    if use_dist:
        # RCCL prints its version banner / warnings with printf on fd 1: keep stdout for the one JSON line
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # UQ_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share cards,
        # the statistics exchange is staged through the host).  The driver's runs use nccl = RCCL, one rank per GPU.
        backend = os.environ.get('UQ_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            local_rank %= max(torch.cuda.device_count(), 1)
    ctx = Context(local_rank)
    from uq_amd.device import SideContext
    side = SideContext(ctx)
    red_dev = ctx.device if not (use_dist and dist.get_backend() == 'gloo') else 'cpu'     # where the closing all-reduces live
    notricks = args.workload == 'cfg5-notricks'
    # The step is the WHOLE encode of the shard, QNAME columns included, at every N.  N > 1: the QNAME layout is a property of the whole
    # file (line 1, the common prefix / suffix): rank 0 guesses it from its reads, ONE broadcast of the structure makes it every rank's,
    # every rank's pack kernel verifies it on all its reads, and flags / value ranges / first occurrences are combined over the ranks
    # (one all-gather of a few integers, one all-reduce of the small first-occurrence tables: qname_device.analyse_fused_sharded).
    qname_in_step = not args.no_qname

    def fence():
        if use_dist: dist.barrier()
        torch.cuda.synchronize()

    def fetch(st):
        if use_dist:
            from uq_amd import dist as uqdist
            return uqdist.allreduce_stats(ctx, st)
        return ops.stats_fetch(ctx, st)

    def run_encode(n, steps, warmup, compare_exact, qname_in_step=qname_in_step):
        """`steps` timed steps of the hot path over n reads per GPU (W untimed first) -> everything the JSON needs."""
        shard = None
        if use_dist:
            from uq_amd import dist as uqdist
            shard = uqdist.Shard(uqdist.HipRows(ctx), rank * n, world * n)       # rank r's shard = reads [r n, (r + 1) n) of the synthetic file
        if args.workload == 'cfg2':
            spec = synth.Spec(SEED, args.length)
        else:
            spec = synth.Spec(20261003 + 5, (36, 301), n_rate=1)
        d_buf = ops.synth_fastq(ctx, spec, rank * n, n)           # resident in HBM before timing
        fastq_bytes = d_buf.numel()
        ctx.sync()
        pack_events = []
        state = {}

        def decide_and_params(hs, nreads):
            d = host_decide(hs, notricks=notricks)
            p = ops.make_pack_params(d['bases'], d['qualities'], d['N_qual'], d['bits_per_base'], d['bits_per_quality'],
                                     d['variable_read_lengths'], d['dna_bytes_per_row'], d['quality_bytes_per_row'],
                                     d['dna_max'], hs.max_record_bytes, avg_record_bytes=fastq_bytes // max(nreads, 1))
            return d, p

        def step(timed):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            spec = None
            guess = None
            census = None
            queued = None
            st = None
            ls_async = None
            fq = None
            if not args.multi_pass:
                # the census kernel is queued first, its closing scan right behind it (the line count stays on the device); the guess
                # (census / index / statistics of the shard's first 4 MB: small kernels, a read-back, the decisions on the host) runs
                # meanwhile on a second stream; the record index and the pack + statistics kernel are then queued behind the census
                # with the count taken on the device -- the host reads it after everything has been queued (no mid-step round trip)
                st_q = ops.stats_new(ctx)                                    # (initialised before the census: off the critical path)
                census = ops.ChunkedCensus(ctx, d_buf)
                census.chunk(0, fastq_bytes)
                census.end_async()
                g = ops.head_guess(side, d_buf, notricks=notricks, head_bytes=ops.HEAD_BYTES_SMALL, head_reads=ops.HEAD_READS_INDEXED)
                if g is None and use_dist and qname_in_step:
                    # N > 1: the broadcast of rank 0's layout guess is a collective -- a rank without a guess of its own takes part all the same
                    # (its flag makes every rank stand down together in analyse_fused_sharded)
                    from uq_amd import qname_device
                    fq = ops.FusedQname(ctx, 16)
                    if rank == 0: fq.q.zero_()                                  # ok = 0: no layout
                    qname_device.broadcast_guess(ctx, fq, shard)
                if g is not None:
                    guess, rpb = g
                    cap_reads = int(fastq_bytes * rpb * 1.02) + 1024
                    guess.avg_record_bytes = int(1.0 / rpb)                     # tile sizing hint: the head's own records
                    # no record index (8 B a line) is expanded: the pack kernel and the QNAME sample walk the census's newline lists
                    # themselves (csrc/lines.h); a fallback that needs the index expands it then (index() below)
                    ls_cap = ls_async = None
                    if qname_in_step:
                        # the QNAME passes ride in the pack kernel: layout guessed on the device from a sample of the reads (queued here,
                        # behind the index), verified on every read while the fields are parsed; distinct counts queued behind it
                        fq = ops.FusedQname(ctx, cap_reads)
                        if rank == 0: ops.qname_guess_async(ctx, d_buf, ls_cap, fq)
                        if use_dist:
                            from uq_amd import qname_device
                            qname_device.broadcast_guess(ctx, fq, shard)             # rank 0's guess, on every rank before its pack kernel
                    e0.record()
                    sp = ops.pack_stats_async(ctx, d_buf, ls_cap, cap_reads, guess, st=st_q, fq=fq)
                    e1.record()
                    if sp is not None and fq is not None: ops.qname_fused_finish(ctx, fq)
                    if sp is not None:
                        # one rank: the statistics' read-back is queued (and waited for) first, the line count is there by then
                        queued = (ls_cap, sp, None if use_dist else fetch(sp[3]))
            hs = None
            if census is not None:
                nlines, ok = census.wait()
                good = queued is not None and ok and nlines % 4 == 0 and nlines // 4 <= cap_reads
                if not ok: nlines = ops.count_lines(ctx, d_buf)         # a tile's newline list overflowed: the bitmap form
                nreads = nlines // 4
                if good:
                    ls = queued[0][:nlines + 1] if queued[0] is not None else None      # None: expanded on demand (index() below)
                    dq = queued[1]
                    spec = (dq[0][:nreads * guess.dna_bytes_per_row], dq[1][:nreads * guess.quality_bytes_per_row], dq[2], dq[3])
                    st = dq[3]
                elif queued is not None:                              # the queued form does not hold for this shard: the plain index, and
                    ls = ops.index_lines(ctx, d_buf, nlines)           # statistics flagged incomplete, so that EVERY rank redoes them below
                    st = queued[1][3]
                    st[ops.STATS_DTYPE.fields['reserved'][1]] = 1     # uq_stats.reserved: statistics incomplete
                    queued = None
                if st is not None:
                    hs = queued[2] if (queued is not None and queued[2] is not None) else fetch(st)     # N > 1: the all-reduce of the statistics
            if hs is None:
                if census is None: nlines = ops.count_lines(ctx, d_buf)
                if census is not None and ls_async is not None and ok and nlines <= 4 * cap_reads:
                    ls = ls_async[:nlines + 1]                        # the queued index holds (there is no fused pack kernel for this alphabet)
                else:
                    ls = ops.index_lines(ctx, d_buf, nlines)          # record index
                nreads = nlines // 4
                st = ops.stats_new(ctx)
                ops.stats_accumulate(ctx, st, d_buf, ls, 0, nreads)
                hs = fetch(st)                                            # N > 1: the all-reduce of the statistics
            def index():                                              # the expanded index, for the kernels that take one (fallbacks)
                return ls if ls is not None else ops.index_lines(ctx, d_buf, nlines)
            if hs.incomplete:                                         # the speculative pass met something outside its guess
                spec = None
                st = ops.stats_new(ctx)
                ls = index()
                ops.stats_accumulate(ctx, st, d_buf, ls, 0, nreads)
                hs = fetch(st)
            if hs.bad_plus is not None or hs.bad_len is not None:
                raise RuntimeError('malformed FASTQ record')
            # the QNAME analysis (uq.py:394-444, 555-678, 717-736): from the pack kernel's QNAME phase when every read conformed to the
            # guessed layout, else by the exact kernels (layout reductions, tokeniser) -- inside the timed step either way.  First: its
            # column encoders run on the device while the host takes the DNA / QUAL decisions below.
            qpath, qres = None, None
            if qname_in_step:
                from uq_amd import qname_device
                if use_dist:                    # collective: a rank whose own pass did not hold says so inside, and every rank stands down with it
                    qres = qname_device.analyse_fused_sharded(ctx, fq, nreads, shard, usable=spec is not None and queued is not None)
                else:
                    qres = qname_device.analyse_fused(ctx, fq, nreads) if (fq is not None and spec is not None) else None
                qpath = 'fused into the pack kernel' + (' (rank 0\'s layout guess broadcast, every rank verifies; flags / ranges / first occurrences combined over the ranks)' if use_dist else '')
                if qres is None:
                    ls = index()
                    qres = qname_device.analyse_device(ctx, d_buf, ls, nreads, shard)
                    qpath = 'exact kernels (layout, tokeniser)' + (' over shards' if use_dist else '')
                if qres is None: raise RuntimeError('the synthetic QNAMEs are outside the device subset')
            d, p = decide_and_params(hs, nreads)
            if spec is not None and ops.same_pack_params(p, guess):
                dna, qual, bad = spec[:3]
                kernel = 'pack_tile_kernel<STATS> (pack + pass-1 statistics in one read of the stream)'
            else:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                ls = index()
                e0.record()
                dna, qual, bad = ops.pack(ctx, d_buf, ls, 0, nreads, p)
                e1.record()
                kernel = 'pack_tile_kernel'
            if timed: pack_events.append((e0, e1, kernel))
            state.update(dna=dna, qual=qual, bad=bad, d=d, nreads=nreads, nlines=nlines, ls=ls, params=p, qname=qres, qname_path=qpath)

        for _ in range(warmup):
            step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        fence()
        dt = time.perf_counter() - t0
        indexed = state['ls'] is not None                                   # did the step expand the record index?
        if state['ls'] is None: state['ls'] = ops.index_lines(ctx, d_buf, ops.count_lines(ctx, d_buf))     # (for the checks below, outside the timed region)
        total_bytes, total_reads = fastq_bytes, state['nreads']
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            tot = torch.tensor([fastq_bytes, state['nreads']], dtype=torch.int64, device=red_dev)   # shards differ by a few bytes
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            total_bytes, total_reads = int(tot[0].item()), int(tot[1].item())
        # For comparison, outside the timed region: the exact QNAME kernels (two traversals of the QNAME lines) on the same shard.
        qname_exact_ms = None
        if compare_exact and not use_dist:
            from uq_amd import qname_device
            best = None
            for _ in range(3):
                q0 = time.perf_counter()
                qex = qname_device.analyse_device(ctx, d_buf, state['ls'], state['nreads'])
                torch.cuda.synchronize()
                q1 = time.perf_counter()
                best = (q1 - q0) if best is None or (q1 - q0) < best else best
                if qex is None: best = None; break
            qname_exact_ms = None if best is None else best * 1e3
            if qname_in_step and qex is not None:
                # the step's columns against the exact path's: same layout, same column descriptions, same arrays
                a, b = state['qname'], qex
                if a[:4] != b[:4] or not all(torch.equal(x, y) for x, y in zip(a[4], b[4])):
                    raise RuntimeError('parity failure: the fused QNAME pass differs from the exact kernels')
        if compare_exact and use_dist and qname_in_step:
            from uq_amd import qname_device
            qex = qname_device.analyse_device(ctx, d_buf, state['ls'], state['nreads'], shard)      # the exact kernels over the same shards (collective)
            a = state['qname']
            if qex is None or a[:4] != qex[:4] or not all(torch.equal(x, y) for x, y in zip(a[4], qex[4])):
                raise RuntimeError('parity failure: the sharded fused QNAME pass differs from the exact sharded kernels')
        if state['bad'] is not None and ops.bad_index(state['bad']) is not None:
            raise RuntimeError('pack reported an uncoded symbol at read %d' % ops.bad_index(state['bad']))
        d, nreads = state['d'], state['nreads']
        kernel = pack_events[-1][2]
        pack_ms = float(np.mean([a.elapsed_time(b) for a, b, k in pack_events if k == kernel]))
        # SURVEY.md 8d bytes per read: the record read once + both rows written (+ the QNAME columns when the kernel writes them:
        # their final width, what an ideal pass would store)
        qcol_bytes = sum(x.element_size() for x in state['qname'][4]) if (state.get('qname') and 'fused' in (state.get('qname_path') or '')) else 0
        algo_bytes = fastq_bytes + nreads * (d['dna_bytes_per_row'] + d['quality_bytes_per_row'] + qcol_bytes)
        step_bytes = fastq_bytes + nreads * (d['dna_bytes_per_row'] + d['quality_bytes_per_row'] + (32 if indexed else 0) + qcol_bytes)
        return dict(indexed=indexed, d_buf=d_buf, fastq_bytes=fastq_bytes, state=state, dt_step=dt / steps, total_bytes=total_bytes, total_reads=total_reads,
                    qname_exact_ms=qname_exact_ms, kernel=kernel, pack_ms=pack_ms, algo_bytes=algo_bytes, step_bytes=step_bytes, d=d, nreads=nreads)

    def roofline_of(m, traffic=None):
        achieved = m['algo_bytes'] / 1e9 / (m['pack_ms'] / 1e3)
        return {'bound': 'hbm', 'kernel': m['kernel'], 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'algorithmic_bytes_per_launch': int(m['algo_bytes']),
                'avg_launch_ms': round(m['pack_ms'], 4)}

    m = run_encode(args.reads, args.steps, args.warmup, True)
    state, d, nreads, fastq_bytes, kernel = m['state'], m['d'], m['nreads'], m['fastq_bytes'], m['kernel']
    two_reads = kernel.startswith('pack_tile_kernel<STATS')

    def measured_traffic(workload, reads, form):
        """HBM bytes per launch of the step's pack kernel from the PMC passes on record (profiles/pack_stats_traffic.json: rocprofv3 --pmc FETCH_SIZE /
        WRITE_SIZE in separate passes, tools/pmc_traffic.sh + tools/update_traffic.py), not from this run: an entry is only quoted while the kernel's
        source is byte for byte the one it was measured on."""
        tpath = os.path.join(HERE, 'profiles', 'pack_stats_traffic.json')
        try:
            import hashlib
            tj = json.load(open(tpath))
            sha = hashlib.sha256(open(os.path.join(HERE, tj['kernel_source']), 'rb').read()).hexdigest()
            for e in tj.get('entries', []):
                if e['kernel_source_sha256'] == sha and e['workload'] == workload and e['reads'] == reads and e['kernel_form'] == form:
                    return e['hbm_bytes_per_launch']
        except Exception:
            pass
        return None

    traffic = measured_traffic(args.workload, nreads, 'qname' if qname_in_step else 'plain') if two_reads and args.length == 150 else None

    mode = ((' [TWO reads of the stream, queued back to back (the line count stays on the device): census' + (' + index' if m['indexed'] else ' (no record index: the kernels walk its newline lists)') +
             ', then pack + statistics in one kernel with decisions guessed from the shard\'s first '
             '8192 reads; the tables were kept because the whole shard\'s statistics gave the same decisions]' if two_reads else
             ' [three reads of the stream: census, statistics, pack]'))
    qmode = ' + QNAME layout / typing / column encoding (%s)' % state.get('qname_path') if qname_in_step else ' (QNAME passes not in the step)'
    result = {
        'metric': 'FASTQ encode MB/s (150bp synthetic; bit-exact tables vs reference)',
        'value': round(m['total_bytes'] / 1e6 / m['dt_step'], 1), 'unit': 'MB/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(m['dt_step'] * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'reads_per_s': round(m['total_reads'] / m['dt_step'], 1),
        'qname': {'in_step': bool(qname_in_step), 'path': state.get('qname_path'),
                  'exact_kernels_ms': None if m['qname_exact_ms'] is None else round(m['qname_exact_ms'], 3)},
        'config': {'workload': ('BASELINE configs[1]: %d x %dbp synth-v1 FASTQ per GPU (%.3f GB), --sort None --raw DNA QUAL QNAME '
                                '--pattern 0.1 0.1; step = census + index + stats + decisions + %d-bit DNA / %d-bit QUAL pack%s%s'
                                % (nreads, args.length, fastq_bytes / 1e9, d['bits_per_base'], d['bits_per_quality'], mode, qmode)) if args.workload == 'cfg2' else
                               ('BASELINE configs[4] (%s): %d x 36-301bp synth-v1 FASTQ with 1%% N per GPU (%.3f GB); step = census + index + '
                                'stats + decisions + %d-bit DNA / %d-bit QUAL variable-length pack%s%s'
                                % (args.workload, nreads, fastq_bytes / 1e9, d['bits_per_base'], d['bits_per_quality'], mode, qmode)),
                   'reads_per_gpu': nreads, 'read_length': args.length if args.workload == 'cfg2' else '36-301', 'fastq_bytes_per_gpu': fastq_bytes,
                   'sharding': 'record-parallel, %d shard(s)' % world},
        'step_roofline': {'algorithmic_bytes_per_step': int(m['step_bytes']),
                          'note': 'one ideal pass: the records read once; both rows and the QNAME columns written' + (', and the 32 B of line offsets per read' if m['indexed'] else
                                  ' (no record index is expanded: the kernels walk the census lists)'),
                          'frac_of_8TBps': round(m['step_bytes'] / 1e9 / m['dt_step'] / HBM_PEAK_GBS, 4)},
        'roofline': roofline_of(m, traffic),
    }

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        ns = min(args.cpu_sample, nreads)
        end = int(state['ls'][4 * ns].item())
        sample = bytes(m['d_buf'][:end].cpu().numpy().tobytes())
        cb, (rd, rq) = cpu_baseline(sample, ns, d)
        gd = state['dna'][:ns * d['dna_bytes_per_row']].cpu().numpy().reshape(ns, -1)
        gq = state['qual'][:ns * d['quality_bytes_per_row']].cpu().numpy().reshape(ns, -1)
        cb['gpu_rows_match_oracle_on_sample'] = bool(np.array_equal(gd, rd) and np.array_equal(gq, rq))
        if not cb['gpu_rows_match_oracle_on_sample']:
            raise RuntimeError('parity failure: GPU rows differ from the oracle on the CPU sample')
        result['cpu_baseline'] = cb
    # the step without the QNAME passes (what the N > 1 lines of rounds 1 - 3 timed), at N = 1: for comparisons with those records
    if qname_in_step and world == 1 and not use_dist:
        plain = run_encode(args.reads, max(5, args.steps // 2), 2, False, qname_in_step=False)
        result['qname']['step_without_qname_ms'] = round(plain['dt_step'] * 1e3, 3)
        result['qname']['value_without_qname'] = round(plain['total_bytes'] / 1e6 / plain['dt_step'], 1)
        del plain
    del m, state
    torch.cuda.empty_cache()

    # ---- the north_star's single-GPU size in the driver's own record: the same step over 200 M x 150 bp (67.9 GB of FASTQ in HBM)
    if world == 1 and not use_dist and args.workload == 'cfg2' and args.north_star_reads > 0 and args.north_star_reads != args.reads:
        big = run_encode(args.north_star_reads, 3, 2, False)       # (two warm-up steps: both generations of the 30 GB of tables are in the allocator's cache)
        result['north_star_200M'] = {
            'workload': 'the same step over %d x %dbp (%.1f GB of FASTQ resident in HBM), 3 timed steps after 2 warm-up steps' % (big['nreads'], args.length, big['fastq_bytes'] / 1e9),
            'ms_per_step': round(big['dt_step'] * 1e3, 3), 'value': round(big['total_bytes'] / 1e6 / big['dt_step'], 1), 'unit': 'MB/s',
            'reads_per_s': round(big['total_reads'] / big['dt_step'], 1), 'qname_path': big['state'].get('qname_path'),
            'step_frac_of_8TBps': round(big['step_bytes'] / 1e9 / big['dt_step'] / HBM_PEAK_GBS, 4),
            'roofline': roofline_of(big, measured_traffic('cfg2', big['nreads'], 'qname' if qname_in_step else 'plain') if args.length == 150 else None)}
        if 'cpu_baseline' in result:
            result['north_star_200M']['times_cpu_baseline'] = round(result['north_star_200M']['value'] / result['cpu_baseline']['value'], 1)
        del big
        torch.cuda.empty_cache()

    # ---- the global --sort legs (the north_star's scaling claim: ">= 6x further at 8 GPUs for --sort DNA"; BASELINE configs[3] is
    # `--sort QUAL --raw DNA QUAL QNAME` over 8 GPUs).  STRONG scaling: args.sort_reads reads over all N GPUs (200 M / N each), so
    # that the N = 1 and N = 8 lines divide directly.  Table movements of the two mixes, through uq_amd.dist (N = 1: the same calls
    # without an exchange):
    #   sort_qual_raw   sample sort of the QUAL rows + the DNA rows moved into that order                 (uq.py:773-777 twice)
    #   sort_dna_keyed  sample sort of the DNA rows + unique / group ids; the same for the QUAL rows + their ids sent back to file
    #                   order and fetched in the DNA order                                                (uq.py:784-798 twice)
    if args.sort_reads > 0 and args.workload == 'cfg2':
        result['sort_leg'] = sort_legs(ctx, args, rank, world, use_dist, fence, fetch, red_dev)
    # (after the sort legs: the CLI's 8 GB of tmpfs files and pinned staging buffers leave the host busy for a while -- the legs' host round trips
    # read 60 ms before it and 78 ms behind it)
    # ---- end to end ("given twice -- kernel and end to end"): the drop-in CLI on a FILE of the same workload, both ways.  File system,
    # PCIe and tar writing are all inside; this is never `value` (which has its input resident in HBM).
    if world == 1 and not use_dist and args.workload == 'cfg2' and not args.no_e2e:
        try:
            result['end_to_end'] = end_to_end(ctx, args)
        except OSError as e:                                   # (no room on the tmpfs of this box: say so instead of failing the line)
            result['end_to_end'] = {'skipped': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    if rank == 0:
        print(json.dumps(result), file=json_out, flush=True)
    if use_dist:
        dist.destroy_process_group()


def end_to_end(ctx, args):
    """BASELINE configs[1] through the CLI (uq_amd.uq.Session, what `python -m uq_amd.uq -i reads.fastq -o reads.uQ --sort None --raw DNA QUAL
    QNAME --pattern 0.1 0.1` runs) from a file on tmpfs to a .uQ on tmpfs, and back to text; the decoded file must be the input."""
    import tempfile
    from uq_amd import ops, synth, uq
    n = args.reads
    d = ops.synth_fastq(ctx, synth.Spec(SEED, args.length), 0, n)
    tmpdir = tempfile.mkdtemp(prefix='uq_e2e_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    path, out, back = os.path.join(tmpdir, 'reads.fastq'), os.path.join(tmpdir, 'reads.uQ'), os.path.join(tmpdir, 'back.fastq')
    try:
        host = ctx.to_numpy(d)
        host.tofile(path)
        nbytes = int(host.size)
        del d
        flags = ['--sort', 'None', '--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '0.1', '0.1']
        a = uq.validate_args(uq.build_parser().parse_args(['-i', path, '-o', out, '--quiet'] + flags))
        s = uq.Session(a, ctx=ctx)
        stages = {}

        def lap(name, fn):
            t = time.perf_counter(); fn(); ctx.sync(); stages[name] = round(time.perf_counter() - t, 4)

        def run():
            lap('file -> pinned -> HBM, census queued per chunk, guess, pack + statistics + QNAME fields', lambda: s.load(path))
            lap('decisions from the whole file, QNAME columns', s.analyse)
            lap('pack (kept: the guess held)', s.pack)
            if a.sort is None: a.sort = (None,)
            if a.raw is None: a.raw = (None,)
            lap('members (--raw: the tables as they are)', lambda: s.run_mix(a.sort, a.raw, False))
            lap('tar: HBM -> pinned -> pwrite', lambda: s.write_container(out))
        t0 = time.perf_counter()
        run()
        enc = time.perf_counter() - t0
        load_path, qpath = s.load_path, s.qname_path
        del s
        a2 = uq.validate_args(uq.build_parser().parse_args(['-i', out, '--decode', '--quiet']))
        t0 = time.perf_counter()
        with open(back, 'wb') as f:
            uq.Session(a2, ctx=ctx).decode(out=f)
        dec = time.perf_counter() - t0
        same = bool(np.array_equal(np.fromfile(back, dtype=np.uint8), host))
        if not same: raise RuntimeError('parity failure: decode(encode(file)) differs from the file')
        return {'workload': 'BASELINE configs[1] through the CLI: %d x %dbp, a %.3f GB file on tmpfs -> .uQ on tmpfs (%.3f GB) -> text again' % (n, args.length, nbytes / 1e9, os.path.getsize(out) / 1e9),
                'encode_s': round(enc, 3), 'decode_s': round(dec, 3), 'MBps': round(nbytes / 1e6 / enc, 1), 'decode_MBps': round(nbytes / 1e6 / dec, 1),
                'stages_s': stages, 'encode_path': load_path, 'qname_path': qpath, 'decoded_file_is_the_input': same,
                'note': 'file system, PCIe and tar writing included; never `value`'}
    finally:
        import shutil
        shutil.rmtree(tmpdir, ignore_errors=True)


def sort_legs(ctx, args, rank, world, use_dist, fence, fetch, red_dev):
    import torch
    import torch.distributed as dist
    from uq_amd import dist as uqdist, ops, synth
    be = uqdist.HipRows(ctx)
    total = args.sort_reads
    lo, hi = uqdist.shard_range(total, rank, world)
    ns = hi - lo
    spec3 = synth.Spec(20261003 + 4, args.length, dup='both', dup_templates=max(1, total // 16))
    buf3 = ops.synth_fastq(ctx, spec3, lo, ns)
    nl3 = ops.count_lines(ctx, buf3)
    ls3 = ops.index_lines(ctx, buf3, nl3)
    st3 = ops.stats_new(ctx)
    ops.stats_accumulate(ctx, st3, buf3, ls3, 0, ns)
    hs3 = fetch(st3)
    d3 = host_decide(hs3)
    p3 = ops.make_pack_params(d3['bases'], d3['qualities'], d3['N_qual'], d3['bits_per_base'], d3['bits_per_quality'], d3['variable_read_lengths'],
                              d3['dna_bytes_per_row'], d3['quality_bytes_per_row'], d3['dna_max'], hs3.max_record_bytes)
    dna3, qual3, _ = ops.pack(ctx, buf3, ls3, 0, ns, p3)
    del buf3, ls3
    torch.cuda.empty_cache()
    Cd3, Cq3 = d3['dna_bytes_per_row'], d3['quality_bytes_per_row']
    starts = [uqdist.shard_range(total, r, world)[0] for r in range(world)] + [total]
    t = torch
    per_rank = [starts[r + 1] - starts[r] for r in range(world)]

    def sort_qual_raw():
        gs = uqdist.global_sort_rows(be, qual3, ns, Cq3, lo, total_rows=total, rows_of_ranks=per_rank)
        dg = uqdist.dist_gather_rows(be, dna3, ns, Cd3, starts, gs['gidx'])
        return gs['rows'], (gs, dg)

    def keyed(table, cols):
        # a keyed table (uq.py:784-789): the global order, the groups the sort's own head flags give, and the distinct rows -- the table member
        return uqdist.global_sort_rows(be, table, ns, cols, lo, total_rows=total, rows_of_ranks=per_rank, want='unique')

    def sort_dna_keyed():
        gd = keyed(dna3, Cd3)
        gq = keyed(qual3, Cq3)
        kd, kq = gd['group'], gq['group']                    # (the per-rank id offsets are a handful of integers: dist_encode._unique)
        in_file_order = uqdist.dist_scatter_rows(be, kq.view(t.uint8), 4, starts, gq['gidx'])
        kq_sorted = uqdist.dist_gather_rows(be, in_file_order, ns, 4, starts, gd['gidx'])
        return gd['rows'], (gd, kd, gq, kq_sorted)

    out = {'workload': 'BASELINE configs[2] / [3] shape: %d x %dbp over %d GPU(s) (%d per GPU, strong scaling), 10 %% of the reads copy one of N/16 '
                       'templates (bases and qualities); rows: DNA %d B, QUAL %d B' % (total, args.length, world, ns, Cd3, Cq3),
           'reads_total': total, 'reads_per_gpu': ns}
    for name, fn, moved in (('sort_qual_raw', sort_qual_raw, Cq3 + 8 + 8 + Cd3), ('sort_dna_keyed', sort_dna_keyed, Cd3 + 8 + Cq3 + 8 + 3 * (4 + 8))):
        rows, keep = fn()
        del keep
        fence()
        ts = time.perf_counter()
        K3 = 2
        for _ in range(K3):
            rows, keep = fn()
            del keep
        fence()
        dts = (time.perf_counter() - ts) / K3
        biggest = rows
        if use_dist:
            t3 = torch.tensor([dts], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
            dts = float(t3.item())
            r3 = torch.tensor([rows], dtype=torch.int64, device=red_dev)
            dist.all_reduce(r3, op=dist.ReduceOp.MAX)
            biggest = int(r3.item())
        out[name] = {'ms': round(dts * 1e3, 3), 'reads_per_s': round(total / dts, 1), 'largest_shard_after_exchange': int(biggest),
                     'exchanged_bytes_per_rank': int(ns * moved * (world - 1) / world)}
        torch.cuda.empty_cache()
    return out


if __name__ == '__main__':
    main()
