#!/bin/bash
# Per-kernel durations of one python tool (rocprofv3 --kernel-trace --stats; the program itself follows `--`).
#   tools/prof_cmd.sh <tag> <script.py> [args...]     -> gpurun_out/<tag>_kernel_stats.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT/$(dirname $tag)
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_$$
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- python3 $R/"$1" "${@:2}" > $OUT/${tag}_prof.log 2>&1
cd $R; f=$(find /tmp/prof_$$ -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python3 profiles/kstats.py $f > $OUT/${tag}_kernel_stats.txt
rm -rf /tmp/prof_$$
tail -4 $OUT/${tag}_prof.log; head -40 $OUT/${tag}_kernel_stats.txt
