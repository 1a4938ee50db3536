#!/bin/bash
# Round-2 closing records -> gpurun_out/r02y/: full GPU suite, bench line (+ rocprofv3 kernel stats of the same command), table kernels at
# BASELINE configs[2] size, decode kernels, CLI end to end.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r02y; mkdir -p $OUT; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q --durations=10 > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/gpu_suite.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.json; echo
python3 bench.py --steps 20 --warmup 5 --one-pass --cpu-sample 0 > $OUT/bench_one_pass.json 2>/dev/null
python3 tools/bench_tables.py --reads 50000000 > $OUT/tables_50M_config3.jsonl 2>/dev/null; echo tables done
python3 tools/bench_tables.py --reads 50000000 --sort QUAL > $OUT/tables_50M_sort_qual.jsonl 2>/dev/null
python3 tools/bench_decode.py > $OUT/decode_kernels.jsonl 2>/dev/null; echo decode done
python3 tools/bench_e2e.py --reads 10000000 --decode > $OUT/cli_e2e.jsonl 2>/dev/null
python3 tools/bench_e2e.py --reads 50000000 --config 3 --flags "--sort DNA" --decode >> $OUT/cli_e2e.jsonl 2>/dev/null; echo e2e done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o p -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $OUT/prof_bench.log 2>&1
cd $R; f=$(find $OUT/prof_bench -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python3 profiles/kstats.py $f > $OUT/bench_kernel_stats.txt
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
head -12 $OUT/bench_kernel_stats.txt
