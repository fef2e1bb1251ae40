#!/bin/bash
# round 5: k_ldpc_wg8.hip built with other AMDGPU scheduler settings (tools/bin/lib_sch*.so, -mllvm ...) against the in-tree build: the headline launch and the short-frame launch, same box
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_sch*.so | sort -V); do
  echo -n "$(basename $lib): "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 9 2>&1 | grep frames | tr '\n' ' '
  DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=16384 timeout 300 python tools/scan_batch.py QPSK-S_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
done; done
