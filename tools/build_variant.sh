#!/bin/bash
# Development: builds tools/bin/lib_<name>.so = libdvbs2hip.so with ONE translation unit recompiled with extra flags
# (CPU container; the .so travels to the GPU box, DVBS2HIP_LIB selects it):  tools/build_variant.sh NAME k_ldpc_wg8 -DSPA_BS=2 ...
set -e
cd "$(dirname "$0")/.."
name=$1; tu=$2; shift 2
python -c "from dvbs2_amd import build; build.build_lib()" > /dev/null
mkdir -p tools/bin
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -c dvbs2_amd/csrc/$tu.hip -o tools/bin/${tu}_$name.o
objs=$(ls dvbs2_amd/lib/*.hip.o | grep -v "/$tu.hip.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/lib_$name.so $objs tools/bin/${tu}_$name.o
echo tools/bin/lib_$name.so
