#!/bin/bash
# tools/ab_pack.sh TAG VARIANT...: the bench step (10 M x 150 bp, QNAME in the step) with each libuqhip variant (uq_amd/_variants/libuqhip_<V>.so;
# `main` = uq_amd/libuqhip.so) on ONE box, twice round-robin -> gpurun_out/<TAG>_ab.txt (ms per step, the pack kernel's average launch)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT/$(dirname $tag)
: > $OUT/${tag}_ab.txt
for round in 1 2; do
  for v in "$@"; do
    lib=$R/uq_amd/_variants/libuqhip_$v.so; [ "$v" = main ] && lib=$R/uq_amd/libuqhip.so
    UQ_LIB_PATH=$lib timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 3 --no-e2e --cpu-sample 0 --north-star-reads 0 --sort-reads 0 > $OUT/${tag}_$v.json 2> $OUT/${tag}_$v.err || { echo "$v FAILED" >> $OUT/${tag}_ab.txt; exit 1; }
    python3 - $OUT/${tag}_$v.json $v $round >> $OUT/${tag}_ab.txt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-12s round %s: step %.3f ms, pack kernel %.4f ms (frac %.4f), without qname %.3f ms' % (sys.argv[2], sys.argv[3], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['qname'].get('step_without_qname_ms', 0)))
PY
  done
done
cat $OUT/${tag}_ab.txt
