#!/bin/bash
# Development: builds tools/bin/lib_<name>.so = libdvbs2hip.so with the listed translation units recompiled with extra flags:
#   tools/build_variant_tus.sh NAME "k_ldpc k_ldpc_cu1" -DLDPC_CU1_HA_V=13 ...      (DVBS2HIP_LIB selects the library; tools/ab_variants.sh)
set -e
cd "$(dirname "$0")/.."
name=$1; tus=$2; shift 2
python -c "from dvbs2_amd import build; build.build_lib()" > /dev/null
mkdir -p tools/bin
objs=$(ls dvbs2_amd/lib/*.hip.o)
for tu in $tus; do
  /opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -c dvbs2_amd/csrc/$tu.hip -o tools/bin/${tu}_$name.o 2>/dev/null &
  objs=$(echo "$objs" | grep -v "/$tu.hip.o")
done; wait
new=""; for tu in $tus; do new="$new tools/bin/${tu}_$name.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/lib_$name.so $objs $new
echo tools/bin/lib_$name.so
