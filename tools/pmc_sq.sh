#!/bin/bash
# Shader-core counters per launch of the kernels of one command, three rocprofv3 --pmc passes of four counters each (with
# --kernel-trace only; the program itself follows `--`).  Millions per launch, summed over the chip.
#   tools/pmc_sq.sh <tag> <kernel regex> <python script> [args...]     -> gpurun_out/<tag>_sq.txt
set -e
tag=$1; pat=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT $(dirname $OUT/$tag)
cd /tmp && export TMPDIR=/tmp
: > $OUT/${tag}_sq.txt
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$i -o p -- python3 $R/"$1" "${@:2}" > $OUT/${tag}_sq_$i.log 2>&1 || { echo "pass $i failed" >> $OUT/${tag}_sq.txt; continue; }
  python3 - /tmp/sq_$i/p_counter_collection.csv "$pat" >> $OUT/${tag}_sq.txt <<'PY'
import collections, csv, re, sys
pat = re.compile(sys.argv[2])
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if not pat.search(k): continue
    m = re.search(r'(\w+_kernel(<\w+>)?)', k); k = m.group(1) if m else k[:50]
    per[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in per: print(k, {c: round(v / len(n[k]) / 1e6, 2) for c, v in sorted(per[k].items())}, '(millions per launch, %d launches)' % len(n[k]))
PY
done
cat $OUT/${tag}_sq.txt
