#!/bin/bash
# Development: builds tools/bin/lib_<name>.so = libdvbs2hip.so with TWO translation units (plan + kernel: k_ldpc, k_ldpc_wg8) recompiled with extra
# flags, e.g. tools/build_variant2.sh nl14 -DDVBS2HIP_PARK_NR=32 -DDVBS2HIP_PARK_NL=14   (DVBS2HIP_LIB selects the library; tools/ab_variants.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python -c "from dvbs2_amd import build; build.build_lib()" > /dev/null
mkdir -p tools/bin
for tu in k_ldpc k_ldpc_wg8; do
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -c dvbs2_amd/csrc/$tu.hip -o tools/bin/${tu}_$name.o 2>/dev/null &
done; wait
objs=$(ls dvbs2_amd/lib/*.hip.o | grep -v "/k_ldpc.hip.o\|/k_ldpc_wg8.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/lib_$name.so $objs tools/bin/k_ldpc_$name.o tools/bin/k_ldpc_wg8_$name.o
echo tools/bin/lib_$name.so
