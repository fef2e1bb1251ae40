#!/bin/bash
# round 5: sum-product kernels on the short frames (and mode 4 on the normal ones), the in-tree library against tools/bin/lib_tev*.so, same box, alternating
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do
  for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_tev*.so 2>/dev/null | sort -V); do
    echo "== $(basename $lib)"; DVBS2HIP_LIB=$PWD/$lib timeout 600 python tools/bench_spa.py 0 16384 3 2>&1 | grep SPA
    DVBS2HIP_LIB=$PWD/$lib DVBS2HIP_LDPC_FAST_MODE=park4 timeout 600 python tools/bench_spa.py 4096 0 3 2>&1 | grep SPA
  done
done
