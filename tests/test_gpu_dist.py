"""GPU, several ranks: the sharded encoder (uq_amd/dist_encode.py) against the oracle's single-process output.
The ranks share the one card of the test box and talk over gloo (device tensors staged through the host);
on an 8-GPU node the same code runs with one rank per GPU over RCCL.  Every `.uQ` member must equal the
oracle's byte for byte -- i.e. the sharded path (file split at record boundaries, all-reduced statistics,
sharded QNAME passes, sample-sort exchange, keys scattered back to file order, pieces written in place)
is invisible in the result."""
import json
import os
import socket
import subprocess
import sys

import pytest

import uq_oracle as O
from uq_amd import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _run_sharded(world, inp, out, flags, backend='gloo', extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), UQ_DIST_BACKEND=backend, PYTHONPATH=REPO, **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, '-m', 'uq_amd.dist_encode', '-i', str(inp), '-o', str(out), '--quiet'] + flags,
                                      env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            raise
        logs.append(o.decode(errors='replace'))
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    return logs


CASES = [
    (2, []),                                                             # unsorted, unique tables + keys in file order
    (3, ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME']),            # BASELINE configs[3] flags
    (2, ['--sort', 'DNA']),                                              # sorted-on table keyed: key in sorted order, others follow
    (3, ['--sort', 'QNAME', '--raw', 'QUAL', '--pattern', '2.2', '1.2']),
    (2, ['--sort', 'QUAL', '--raw', 'QNAME', '--pattern', '3.1', '0.2']),
    (3, ['--raw', 'DNA', 'QUAL', 'QNAME', '--pattern', '1.1', '3.2', '--notricks']),
]


def test_sharded_encoder_on_rccl_single_rank(tmp_path):
    """The same program over the nccl (= RCCL) backend; one rank is all a one-GPU box can host."""
    _check(tmp_path, 1, ['--sort', 'QUAL', '--raw', 'DNA'], backend='nccl')


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_encoder_runs_the_benched_step(tmp_path, world):
    """Each rank runs the single-GPU step on its shard -- census, rank 0's QNAME guess (broadcast), ONE pack + statistics + QNAME
    field pass without a record index -- and the column analysis comes out of the fused pass's all-gathered facts; the exact
    sharded kernels are the fallback, not the default.  The container is the oracle's."""
    fq = synth.fastq(20261005, 45000 * world, 36, n_rate=2)                          # rank 0 holds the checkpoints at 10 000 .. 40 000 reads
    logs = _check(tmp_path, world, ['--sort', 'QUAL'], fq=fq, extra_env={'UQ_TIMING': '1'})
    line = [l for l in '\n'.join(logs).splitlines() if '"uq_timing": "dist_encode"' in l]
    assert len(line) == 1, logs
    rec = json.loads(line[0])
    assert rec['qname'] == 'fused' and 'no record index' in rec['load'], rec


@pytest.mark.parametrize('world,flags', CASES, ids=lambda v: str(v).replace(' ', ''))
def test_sharded_encode_equals_oracle(tmp_path, world, flags):
    _check(tmp_path, world, flags)


@pytest.mark.parametrize('world,flags', [(3, ['--sort', 'DNA']), (4, ['--sort', 'QUAL']), (3, ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME'])],
                         ids=lambda v: str(v).replace(' ', ''))
def test_sharded_encode_with_a_heavy_tie_group(tmp_path, world, flags):
    """Sixty per cent of the reads are copies of ONE read: the value takes up several splitters of the sample sort, its rows are dealt
    over those ranks by file position, and the unique tables / keys are stitched at the rank boundaries -- the container is still
    the oracle's, byte for byte."""
    import numpy as np
    base = synth.fastq(20261003 + 47, 2400, 40, n_rate=1)
    recs = base.split(b'\n')
    recs = [b'\n'.join(recs[4 * i:4 * i + 4]) + b'\n' for i in range(2400)]
    rng = np.random.default_rng(5)
    twin = recs[7].split(b'\n')
    out = []
    for i, r in enumerate(recs):
        if rng.random() < 0.6:
            q = r.split(b'\n')
            r = q[0] + b'\n' + twin[1] + b'\n+\n' + twin[3] + b'\n'                  # this read's name, the twin's bases and qualities
        out.append(r)
    _check(tmp_path, world, flags, fq=b''.join(out))


@pytest.mark.parametrize('world', [3, 4])
def test_sharded_encode_with_heavy_qname_strings(tmp_path, world):
    """a QNAME string field heavier than a rank's share (a filter flag that is 'N' for 90 % of the reads, a constant word in
    mid-name) is dealt over several ranks by the sort of the field keys; the distinct counts and the sorted map must count it once."""
    from test_gpu_e2e import _records
    base = _records(synth.fastq(20261003 + 48, 3000, 30, n_rate=1))
    name = lambda i: b'@q:%d:%s:lane:%s:%d' % (i % 40, b'Y' if i % 10 == 3 else b'N', b'x' if i < 2990 else b'w', i % 9)
    fq = b''.join(name(i) + b'\n' + b'\n'.join(r.split(b'\n')[1:]) + b'\n' for i, r in enumerate(base))
    _check(tmp_path, world, ['--raw', 'DNA', 'QUAL', 'QNAME'], fq=fq)
    _check(tmp_path, world, ['--sort', 'QNAME'], fq=fq)


def test_sharded_encode_with_idle_ranks(tmp_path):
    """Two reads over three ranks: a rank without reads still takes part in every exchange."""
    fq = b'@a:1:7\nACGTN\n+\nIHIH#\n@a:2:9\nACGTA\n+\nHIHII\n'
    _check(tmp_path, 3, ['--sort', 'DNA'], fq=fq)
    _check(tmp_path, 3, ['--raw', 'DNA', 'QUAL', 'QNAME'], fq=fq)


@pytest.mark.parametrize('seed', [1000, 1007, 1013, 1021, 1034, 1046] + list(range(int(os.environ.get('UQ_DIST_FUZZ_FROM', '5000')), int(os.environ.get('UQ_DIST_FUZZ_FROM', '5000')) + int(os.environ.get('UQ_DIST_FUZZ_N', '0')))))      # UQ_DIST_FUZZ_N=100: a longer hunt
def test_sharded_encode_fuzz(tmp_path, seed):
    """Random alphabets / widths / lengths / flag mixes (the generator of test_gpu_e2e's CLI fuzz) through 2 or 3 ranks;
    tiny files leave ranks without reads."""
    import numpy as np
    from test_gpu_e2e import _fuzz_case, _oracle_flags
    fq, flags = _fuzz_case(np.random.default_rng(seed))
    import re
    try:
        O.encode(fq, **_oracle_flags(flags))
    except (O.UqError, ValueError, IndexError, re.error):
        pytest.skip('the reference refuses this input')
    _check(tmp_path, 2 + seed % (2 if seed < 5000 else 3), flags, fq=fq)


def _check(tmp_path, world, flags, backend='gloo', fq=None, extra_env=None):
    if fq is None: fq = synth.fastq(20261003 + 40, 3000, (30, 61), n_rate=2, dup='both', dup_templates=40)
    inp = tmp_path / 'in.fastq'; inp.write_bytes(fq)
    out = tmp_path / 'out.uQ'
    logs = _run_sharded(world, inp, out, flags, backend, extra_env)
    cfg, members = O.read_tar(str(out))

    from test_gpu_e2e import _oracle_flags
    of = _oracle_flags(flags)
    ocfg, omembers, _ = O.encode(fq, **of)
    assert set(members) == set(omembers)
    for k in omembers:
        assert members[k] == omembers[k], k
    for k in ocfg:
        if k in ('sort', 'raw', 'pattern'): continue
        assert json.loads(json.dumps(cfg[k])) == json.loads(json.dumps(ocfg[k])), k
    new_n_code = ocfg['N_qual'] and max(ocfg['N_qual'].values()) >= len(ocfg['qualities'])      # Q9: not decodable by the reference either
    assert of['sort'] is not None or new_n_code or O.decode(cfg, members) == fq.decode('latin-1')
    return logs


DECODE_CASES = [
    (2, []),                                                             # keys + unique tables: key slices, whole tables
    (3, ['--sort', 'QUAL', '--raw', 'DNA', 'QUAL', 'QNAME']),            # raw row slices read straight from the file
    (2, ['--sort', 'DNA', '--pattern', '2.2', '1.2']),                   # column-major payloads: loaded whole, sliced after
    (3, ['--raw', 'DNA', 'QNAME', '--pattern', '3.1', '0.2', '--notricks']),
    (5, ['--sort', 'QNAME']),
]


@pytest.mark.parametrize('world,flags', DECODE_CASES, ids=lambda v: str(v).replace(' ', ''))
def test_sharded_decode_equals_oracle_decode(tmp_path, world, flags):
    """`--decode` over several ranks: the file the ranks write together is the oracle's decode of the same container
    (and, unsorted, the input)."""
    fq = synth.fastq(20261003 + 41, 2500, (30, 61), n_rate=2, dup='both', dup_templates=40)
    inp = tmp_path / 'in.fastq'; inp.write_bytes(fq)
    enc = tmp_path / 'out.uQ'
    _run_sharded(1, inp, enc, flags)                                    # written by the same program on one rank
    cfg, members = O.read_tar(str(enc))
    out = tmp_path / 'back.fastq'
    _run_sharded(world, enc, out, ['--decode'])
    assert out.read_bytes().decode('latin-1') == O.decode(cfg, members)
    if '--sort' not in flags: assert out.read_bytes() == fq


def test_sharded_decode_with_idle_ranks(tmp_path):
    fq = b'@a:1:7\nACGTN\n+\nIHIH#\n@a:2:9\nACGTA\n+\nHIHII\n'
    inp = tmp_path / 'in.fastq'; inp.write_bytes(fq)
    enc = tmp_path / 'out.uQ'
    _run_sharded(1, inp, enc, ['--raw', 'DNA', 'QUAL', 'QNAME'])
    out = tmp_path / 'back.fastq'
    _run_sharded(3, enc, out, ['--decode'])
    assert out.read_bytes() == fq


def _fused_qname_over_ranks(tmp_path, world, names, tag):
    recs = _records_of(len(names))
    fq = b''.join(nm + b'\n' + b'\n'.join(r.split(b'\n')[1:]) + b'\n' for nm, r in zip(names, recs))
    p = tmp_path / ('%s.fastq' % tag)
    p.write_bytes(fq)
    env = dict(os.environ, PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port()), os.path.join(REPO, 'tests', 'sharded_fused_qname_job.py'), str(p)],
                         env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode(errors='replace')
    assert out.returncode == 0, text[-3000:]
    return [l for l in text.splitlines() if l.startswith('OK') or l.startswith('DECLINED')][-1]


_RECS = {}
def _records_of(n):
    from test_gpu_e2e import _records
    if n not in _RECS: _RECS[n] = _records(synth.fastq(20261003 + 49, n, 24, n_rate=1))
    return _RECS[n]


@pytest.mark.parametrize('world', [2, 3])
def test_fused_qname_pass_over_shards(tmp_path, world):
    """The QNAME passes of the N > 1 step (bench.py; qname_device.analyse_fused_sharded): rank 0's layout guess broadcast, every rank's pack
    kernel verifies and parses, flags / value ranges / first occurrences combined over the ranks.  Illumina-like names over 150 000 reads
    (checkpoints at 10 000 ... 80 000 and the last read fall into different shards; a small-range column whose distinct values keep growing
    across the shards -- a mapping turned integers only by the FILE's counts --, flow-cell coordinates that fire at rank 0's first
    checkpoint, a column whose value range only the last rank completes): the concatenated columns are the oracle's.  (A field that is
    constant on rank 0 and varies later would make rank 0's guess take it into the suffix: the other ranks' flags then stand the pass down.)  A name that breaks the layout on the last
    rank only, and a column that stays a mapping of numbers: every rank stands down together."""
    n = 150_000
    names = [b'@M01:7:FC:%d:%d:%d:%d:%d' % (1 + i % 4, 1101 + (i // 40) % 3000, 1000 + (i * 7919) % 28000, 1000 + (i * 104729) % 28000, 1 + (i // 25000)) for i in range(n)]
    assert _fused_qname_over_ranks(tmp_path, world, names, 'ok').startswith('OK 5')
    broken = list(names); broken[n - 7] = b'@M01:7:FC:2:1101:x1:2:1'
    assert _fused_qname_over_ranks(tmp_path, world, broken, 'broken') == 'DECLINED'
    few = [b'@q:%d:%d' % ([5, 70000, 12345678, 31][i % 4], i % 3) for i in range(50_000)]          # four far-apart numbers: a mapping of strings
    assert _fused_qname_over_ranks(tmp_path, world, few, 'mapping') == 'DECLINED'


def test_bench_two_ranks_rehearsal():
    """bench.py's N > 1 path (sharded synthetic input, all-reduced statistics, MAX-over-ranks timing, one JSON line from
    rank 0) with two ranks sharing the card over gloo; the driver runs the same code over RCCL, one rank per GPU."""
    env = dict(os.environ, UQ_DIST_BACKEND='gloo', PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                          '--reads', '200000', '--sort-reads', '150000'], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode(errors='replace')[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak' and r['value'] > 0 and r['config']['reads_per_gpu'] == 200000
    assert r['roofline']['bound'] == 'hbm' and 'cpu_baseline' not in r          # the CPU baseline is timed at N = 1 only
    assert r['qname']['in_step'] is True and r['qname']['path'].startswith('fused into the pack kernel')   # rank 0's guess broadcast, all ranks verify: the whole encode at N > 1 too (checked in the run against the exact sharded kernels)
    # the global --sort legs, strong-scaled (150 000 reads over the two ranks): sample sort + all-to-all(v) of the QUAL rows with the
    # DNA rows moved along; and the keyed --sort DNA mix (two sorts, group ids scattered back and fetched in the DNA order)
    sl = r['sort_leg']
    assert sl['reads_total'] == 150000 and sl['reads_per_gpu'] == 75000
    for leg in ('sort_qual_raw', 'sort_dna_keyed'):
        assert sl[leg]['ms'] > 0 and 75000 <= sl[leg]['largest_shard_after_exchange'] <= 100000, sl[leg]


def _run_sharded_expect_error(world, inp, out, flags, timeout=120):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), UQ_DIST_BACKEND='gloo', PYTHONPATH=REPO)
        procs.append(subprocess.Popen([sys.executable, '-m', 'uq_amd.dist_encode', '-i', str(inp), '-o', str(out), '--quiet'] + flags,
                                      env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            raise AssertionError('a rank was left waiting in a collective')
        logs.append(o.decode(errors='replace'))
    assert all(p.returncode == 1 for p in procs), '\n'.join(logs)        # every rank leaves with the CLI's error code
    assert not os.path.exists(str(out)) and not os.path.exists(str(out) + '.part')
    return logs[0]


def test_sharded_encoder_errors_reach_every_rank(tmp_path):
    """Malformed input seen by one rank only: all ranks exit with code 1 well inside the timeout and rank 0 prints what
    the single-GPU CLI prints for the same file."""
    from uq_amd import uq
    fq = synth.fastq(20261003 + 45, 600, (30, 61))
    recs = fq.split(b'\n')
    inp, out = tmp_path / 'in.fastq', tmp_path / 'out.uQ'

    def single(data):
        inp.write_bytes(data)
        args = uq.validate_args(uq.build_parser().parse_args(['-i', str(inp), '-o', str(tmp_path / 's.uQ'), '--quiet']))
        from uq_amd.device import Context
        c = Context(0)
        try:
            with pytest.raises(uq.UqError) as e:
                uq.Session(args, ctx=c).encode()
        finally:
            c.close()
        return str(e.value)

    # no final newline
    want = single(fq[:-1])
    assert 'not divisible by 4' in want
    assert want in _run_sharded_expect_error(2, inp, out, [])
    # third line of a record in the SECOND half does not start with '+'
    bad = list(recs); bad[4 * 450 + 2] = b'-'
    want = single(b'\n'.join(bad))
    assert 'the third line does not start with +' in want
    assert want in _run_sharded_expect_error(3, inp, out, ['--sort', 'DNA'])
    # SEQ / QUAL lengths differ in the last record
    bad = list(recs); bad[4 * 599 + 3] = bad[4 * 599 + 3][:-2]
    want = single(b'\n'.join(bad))
    assert 'does not match the length of the quality scores' in want
    assert want in _run_sharded_expect_error(2, inp, out, [])
