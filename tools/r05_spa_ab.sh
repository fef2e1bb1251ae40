#!/bin/bash
# round 5: variants of the sum-product mode-6 kernel (tools/bin/lib_spa*.so) against the in-tree library, DVBS2HIP_LDPC_FAST_MODE=cu1, same box, alternating
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do
  for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_spa*.so 2>/dev/null | sort -V); do
    echo -n "$(basename $lib) "; DVBS2HIP_LIB=$PWD/$lib DVBS2HIP_LDPC_FAST_MODE=cu1 timeout 600 python tools/bench_spa.py 16384 0 3 2>&1 | grep SPA
  done
done
