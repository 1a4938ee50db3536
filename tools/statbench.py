import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from uq_amd import ops, synth
from uq_amd.device import Context
ctx = Context(0)
n = 10_000_000
d_buf = ops.synth_fastq(ctx, synth.Spec(20261005, 150), 0, n)
nl = ops.count_lines(ctx, d_buf); ls = ops.index_lines(ctx, d_buf, nl)
for it in range(3):
    st = ops.stats_new(ctx)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.stats_accumulate(ctx, st, d_buf, ls, 0, n); e1.record(); torch.cuda.synchronize()
print('mode', os.environ.get('UQ_STATS_MODE'), 'stats ms', e0.elapsed_time(e1))
