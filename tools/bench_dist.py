#!/usr/bin/env python3
"""Times the sharded encoder (uq_amd/dist_encode.py) on a synth-v1 FASTQ in /dev/shm and checks its output against the
single-GPU CLI's, member by member.  On a one-GPU box the ranks share the card over gloo (host-staged exchange: the
timing of W > 1 is a functional rehearsal, not a scaling number).

    python tools/bench_dist.py [--reads 10000000] [--world 2] [--backend gloo] [--flags "--sort QUAL --raw DNA QUAL QNAME"]
"""
import argparse, hashlib, json, os, socket, subprocess, sys, tarfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def members_sha(path):
    out = {}
    with tarfile.open(path) as t:
        for m in t.getmembers():
            h = hashlib.sha256(); f = t.extractfile(m)
            while True:
                b = f.read(1 << 24)
                if not b: break
                h.update(b)
            out[m.name] = h.hexdigest()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=10_000_000)
    ap.add_argument('--length', type=int, default=150)
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--backend', default='gloo')
    ap.add_argument('--flags', default='--sort QUAL --raw DNA QUAL QNAME')
    a = ap.parse_args()
    path = '/dev/shm/uq_dist_%d.fastq' % os.getpid()
    gen = ("import sys; sys.path.insert(0, %r)\nfrom uq_amd import ops, synth\nfrom uq_amd.device import Context\nctx = Context(0)\n"
           "d = ops.synth_fastq(ctx, synth.Spec(20261003 + 4, %d, dup='qual', dup_templates=%d), 0, %d)\nctx.to_numpy(d).tofile(%r)\n"
           % (REPO, a.length, max(1, a.reads // 16), a.reads, path))
    subprocess.run([sys.executable, '-c', gen], check=True)
    try:
        flags = a.flags.split()
        def work_s(err):
            for line in err.decode(errors='replace').splitlines():
                if line.startswith('{"uq_timing"'): return json.loads(line)['work_s']
        t0 = time.perf_counter()
        r1 = subprocess.run([sys.executable, '-m', 'uq_amd.uq', '-i', path, '-o', path + '.one.uQ', '--quiet'] + flags, check=True, cwd=REPO,
                            env=dict(os.environ, UQ_TIMING='1'), stderr=subprocess.PIPE)
        t_one = time.perf_counter() - t0
        w_one = work_s(r1.stderr)
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, '-m', 'uq_amd.dist_encode', '-i', path, '-o', path + '.dist.uQ', '--quiet'] + flags, cwd=REPO,
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(a.world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                                           MASTER_PORT=str(port), UQ_DIST_BACKEND=a.backend, UQ_TIMING='1'), stderr=subprocess.PIPE if r == 0 else None)
                 for r in range(a.world)]
        err0 = procs[0].communicate()[1]
        rc = [p.wait() for p in procs]
        t_dist = time.perf_counter() - t0
        w_dist = work_s(err0)
        same = rc == [0] * a.world and members_sha(path + '.one.uQ') == members_sha(path + '.dist.uQ')
        print(json.dumps({'op': 'sharded_encode', 'reads': a.reads, 'flags': a.flags, 'world': a.world, 'backend': a.backend,
                          'fastq_MB': round(os.path.getsize(path) / 1e6, 1), 'single_gpu_cli_s': round(t_one, 2),
                          'sharded_cli_s': round(t_dist, 2), 'single_gpu_work_s': w_one, 'sharded_work_s': w_dist,
                          'note': '*_cli_s = whole process (interpreter + torch import + device / process-group start-up); *_work_s = the encode itself, '
                                  'timed inside the process once that is up', 'members_identical': bool(same)}), flush=True)
    finally:
        for p in (path, path + '.one.uQ', path + '.dist.uQ', path + '.dist.uQ.part'):
            if os.path.exists(p): os.remove(p)


if __name__ == '__main__':
    main()
