#!/bin/bash
# The round's standing evidence for the bench kernel at the CURRENT pack.hip, one call: PMC traffic (10 M, 200 M, configs[4] both modes at 100 M), shader-core
# counters and per-kernel durations of the 10 M step.   tools/round_evidence.sh <tag>   -> gpurun_out/<tag>/...; then tools/update_traffic.py per entry (here)
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$tag; mkdir -p $OUT
K='pack_tile_kernel'
B="--cpu-sample 0 --north-star-reads 0 --sort-reads 0 --no-e2e"
set -e
tools/pmc_traffic.sh $tag/bench_10M "$K" bench.py $B --steps 3 --warmup 1 > $OUT/a.log 2>&1; echo "10M traffic done"
tools/pmc_sq.sh $tag/bench_10M "$K" bench.py $B --steps 3 --warmup 1 > $OUT/b.log 2>&1; echo "10M sq done"
tools/prof_bench.sh $tag/bench_10M $B --steps 10 --warmup 2 > $OUT/c.log 2>&1; echo "10M kernel stats done"
tools/pmc_traffic.sh $tag/bench_200M "$K" bench.py --reads 200000000 $B --steps 2 --warmup 1 > $OUT/d.log 2>&1; echo "200M traffic done"
tools/pmc_traffic.sh $tag/cfg5_ntrick_100M "$K" bench.py --workload cfg5-ntrick --reads 100000000 $B --steps 2 --warmup 1 > $OUT/e.log 2>&1; echo "cfg5 ntrick traffic done"
tools/pmc_traffic.sh $tag/cfg5_notricks_100M "$K" bench.py --workload cfg5-notricks --reads 100000000 $B --steps 2 --warmup 1 > $OUT/f.log 2>&1; echo "cfg5 notricks traffic done"
ls $OUT
