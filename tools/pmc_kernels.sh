#!/bin/bash
# SQ counters of the kernels of one command, per launch (rocprofv3 --pmc, one counter set per pass; no other tracing).
#   tools/pmc_kernels.sh <kernel-name-substrings, comma separated> -- python3 tools/bench_decode.py --reps 1
set -e
names="$1"; shift; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- "$@" > /dev/null 2>&1
  NAMES="$names" python3 - <<'PY'
import csv, collections, os
keys = os.environ['NAMES'].split(',')
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open('/tmp/pmc/p_counter_collection.csv')):
    k = r['Kernel_Name']
    for key in keys:
        if key in k:
            acc[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
for name in acc:
    print(name, {c: round(v / cnt[(name, c)] / 1e6, 2) for c, v in acc[name].items()}, '(millions per launch)')
PY
done
