#!/bin/bash
# round 5: the sum-product decoder on normal frames, mode 6 (one frame per CU, two lanes per check: DVBS2HIP_LDPC_FAST_MODE=cu1) against the default (mode 4), same box
cd "${GRAFT_REPO_ROOT:-.}"
if [ "${R05_TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu -k "image_modes and SPA" 2>&1 | tail -5
fi
for i in 1 2 3; do
  for mode in "" cu1; do
    for F in 4096 16384; do
      if [ -z "$mode" ]; then echo -n "default "; timeout 600 python tools/bench_spa.py $F 0 3 2>&1 | grep SPA
      else echo -n "mode=$mode "; DVBS2HIP_LDPC_FAST_MODE=$mode timeout 600 python tools/bench_spa.py $F 0 3 2>&1 | grep SPA; fi
    done
  done
done
