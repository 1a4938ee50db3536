#!/bin/bash
# Round-2 north-star-size records: bench.py at 200 M x 150 bp (single GPU) and configs[4] at 100 M, each once plain
# (the JSON line) and once under rocprofv3 --kernel-trace --stats (per-kernel durations).  Writes gpurun_out/r02b/.
set -e
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r02b
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$PWD}
python3 bench.py --reads 200000000 --steps 5 --warmup 1 > $OUT/bench_200M.json 2> $OUT/bench_200M.err
echo "200M done"; tail -c 600 $OUT/bench_200M.json
python3 bench.py --workload cfg5-ntrick --reads 100000000 --steps 5 --warmup 1 --cpu-sample 20000 > $OUT/bench_cfg5_ntrick_100M.json 2> $OUT/bench_cfg5_ntrick.err
python3 bench.py --workload cfg5-notricks --reads 100000000 --steps 5 --warmup 1 --cpu-sample 20000 > $OUT/bench_cfg5_notricks_100M.json 2> $OUT/bench_cfg5_notricks.err
echo "cfg5 done"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_200M -o p -- python3 $R/bench.py --reads 200000000 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/prof_200M.log 2>&1
echo "prof 200M done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cfg5 -o p -- python3 $R/bench.py --workload cfg5-ntrick --reads 100000000 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/prof_cfg5.log 2>&1
echo "prof cfg5 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_10M -o p -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $OUT/prof_10M.log 2>&1
cd $R
for d in prof_200M prof_cfg5 prof_10M; do f=$(find $OUT/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python3 profiles/kstats.py $f > $OUT/$d.kstats.txt; done
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
ls -la $OUT
