#!/bin/bash
# round 5: sum-product kernel on the short frames with the 16-bit per-lane address table (tools/bin/lib_at16on.so: -DLDPC_SPA_AT16=1 on k_ldpc.hip and k_ldpc_wg8.hip) against the
# in-tree form that builds the addresses on the vector ALU, same box, alternating; then the sum-product parity tests on the variant
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do
  for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_at16*.so 2>/dev/null | sort -V); do
    echo "== $(basename $lib)"; DVBS2HIP_LIB=$PWD/$lib timeout 600 python tools/bench_spa.py 0 16384 3 2>&1 | grep "SPA"
  done
done
DVBS2HIP_LIB=$PWD/tools/bin/lib_at16on.so timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu -k "spa or SPA" 2>&1 | tail -3
