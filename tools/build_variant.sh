#!/bin/bash
# development: build/variants/<name>.so = the product library with ONE source file (default beam_exact.hip) compiled with
# -DJAMD_DEV and extra flags.   usage: [SRC=julius_amd/csrc/dnn.hip] tools/build_variant.sh name -DJAMD_XBEAM_PROBE=1 ...
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/variants
CS=julius_amd/csrc
SRC=${SRC:-$CS/beam_exact.hip}
base=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Wall -Wno-unused-function \
  -DJAMD_DEV "$@" -I$CS -c $SRC -o build/variants/${base}_$name.o
objs=$(ls $CS/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/$name.so $objs build/variants/${base}_$name.o
rm -f build/variants/${base}_$name.o
echo built build/variants/$name.so
