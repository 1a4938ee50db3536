#!/bin/bash
# Start / duration of every kernel of ONE bench step (rocprofv3 --kernel-trace; the program itself follows `--`): where the device
# idles between launches.   [TL_MATCH=kernel] tools/step_timeline.sh <tag> [bench.py args...]   -> gpurun_out/<tag>_timeline.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT/$(dirname $tag)
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tl_$$
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$$ -o p -- python3 $R/bench.py "$@" > $OUT/${tag}_tl.log 2>&1
f=$(find /tmp/tl_$$ -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/${tag}_timeline.txt <<'PY'
import csv, re, sys
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
def short(k):
    m = re.search(r'(\w+_kernel)', k); return m.group(1) if m else k[:40]
# steps start at the big census launches (the small one of the head guess is < 0.3 ms)
starts = [i for i, (s, e, k) in enumerate(rows) if 'census_list_kernel' in k and e - s > 300000]
if len(starts) < 4: sys.exit('no steps found')
a, b = starts[-3], starts[-2]                      # a late, warmed-up step
import os
want = os.environ.get('TL_MATCH')                  # TL_MATCH=<kernel name>: the latest step that runs it (bench.py times several kinds of step)
if want:
    for i in range(len(starts) - 2, 0, -1):
        if any(want in k for s, e, k in rows[starts[i - 1]:starts[i]]) and any(want in k for s, e, k in rows[starts[i]:starts[i + 1]]):
            a, b = starts[i - 1], starts[i]; break
t0 = rows[a][0]; prev_end = t0
print('one step: %.3f ms from census start to the next census start' % ((rows[b][0] - t0) / 1e6))
for s, e, k in rows[a:b]:
    print('%9.1f us  +%7.1f us  (idle before: %6.1f)  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(k)))
    prev_end = max(prev_end, e)
print('%9.1f us  next step starts (idle before: %.1f)' % ((rows[b][0] - t0) / 1e3, (rows[b][0] - prev_end) / 1e3))
PY
rm -rf /tmp/tl_$$; cat $OUT/${tag}_timeline.txt
