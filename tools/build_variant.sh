#!/bin/bash
# development: build/variants/<name>.so = the product library with beam_exact.hip compiled with extra flags
# usage: tools/build_variant.sh name -DJAMD_XBEAM_PROBE=1 ...
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/variants
CS=julius_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Wall -Wno-unused-function \
  -DJAMD_DEV "$@" -I$CS -c ${SRC:-$CS/beam_exact.hip} -o build/variants/beam_exact_$name.o
objs=$(ls $CS/*.o | grep -v beam_exact.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so $objs build/variants/beam_exact_$name.o
echo built build/variants/$name.so
