#!/bin/bash
# HBM traffic per launch of the kernels of one command: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes
# (each with --kernel-trace only; the program itself follows `--`), corrected by profiles/traffic.py as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2 for wide streaming reads).
#   tools/pmc_traffic.sh <tag> <kernel regex> <python script> [args...]     -> gpurun_out/<tag>_traffic.json (+ the two counter csv files)
set -e
tag=$1; pat=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT $(dirname $OUT/$tag)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/"$1" "${@:2}" > $OUT/${tag}_pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $OUT/${tag}_pmc_$(echo $c | tr A-Z a-z).csv
done
cd $R
# the per-dispatch csv files are large: keep only the rows of the kernels asked for (the regex sees the whole kernel name, template
# arguments included: one instance of a kernel can be told from the others), then sum per kernel
for c in fetch_size write_size; do head -1 $OUT/${tag}_pmc_$c.csv > $OUT/t.csv; grep -E "$pat" $OUT/${tag}_pmc_$c.csv >> $OUT/t.csv || true; mv $OUT/t.csv $OUT/${tag}_pmc_$c.csv; done
python3 profiles/traffic.py $OUT/${tag}_pmc_fetch_size.csv $OUT/${tag}_pmc_write_size.csv "" $OUT/${tag}_traffic.json
