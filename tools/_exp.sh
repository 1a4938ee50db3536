#!/bin/bash
# temporary: decode variants
cd "$(dirname "$0")/.."
A="--only-fused --reps 7 --var-min 36 --length 301 --n-rate 1"
python tools/bench_decode.py $A 2>&1 | grep decode_fastq | sed "s/^/base /"
for e in "$@"; do
  UQ_LIB_PATH=$PWD/uq_amd/_variants/libuqhip_$e.so python tools/bench_decode.py $A --no-check 2>&1 | grep decode_fastq | sed "s/^/$e /"
done
