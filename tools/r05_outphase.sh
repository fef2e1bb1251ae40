#!/bin/bash
# round 5: where the output phase's time goes (-DLDPC_PHASE_PROF -DLDPC_PROF_OUT variants under tools/bin), bits socket alone and fused chain
cd "${GRAFT_REPO_ROOT:-.}"
for lib in $(ls tools/bin/lib_po*.so | sort -V); do
  echo "=== $lib  bits socket"
  DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 2 2>&1 | grep -v amdgpu.ids | tail -14
  echo "=== $lib  chain"
  DVBS2HIP_LIB=$PWD/$lib timeout 300 python tools/chain_time.py 2>&1 | grep -v amdgpu.ids | tail -15
done
