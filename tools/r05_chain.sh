#!/bin/bash
# round 5, item 1: the fused chain with the BCH verification inside the LDPC kernel (default) against round 4's form (DVBS2HIP_CHAIN_SYN=0), same box, alternating
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
if [ "${R05_TESTS:-1}" = 1 ]; then
  timeout 1500 python -m pytest tests/test_chain_gpu.py tests/test_golden_gpu.py tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -5
fi
for i in 1 2 3; do
  for syn in 0 1; do
    echo -n "SYN=$syn socket: "; SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
    echo -n "SYN=$syn "; DVBS2HIP_CHAIN_SYN=$syn timeout 300 python tools/chain_time.py 2>&1 | tail -1
    echo -n "SYN=$syn "; DVBS2HIP_CHAIN_SYN=$syn timeout 300 python tools/chain_time.py 16APSK-N_8/9 4096 20 2>&1 | tail -1
  done
done
