#!/bin/bash
# tools/variants.sh FILE.hip NAME FLAGS... : builds uq_amd/_variants/libuqhip_NAME.so = the library with FILE.hip compiled with the extra FLAGS (A/B timing: UQ_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
f=$1; name=$2; shift 2
mkdir -p uq_amd/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c uq_amd/csrc/$f -o uq_amd/_variants/${f%.hip}_$name.o
objs=$(ls uq_amd/csrc/_obj/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o uq_amd/_variants/libuqhip_$name.so $objs uq_amd/_variants/${f%.hip}_$name.o
echo uq_amd/_variants/libuqhip_$name.so
