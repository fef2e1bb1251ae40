#!/bin/bash
# same-box A/B of library variants under tools/bin (DVBS2HIP_LIB) against the in-tree library: parity test first, then the bench-sized launch, alternating
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
TESTK="${TESTK:-image_modes and NMS and not cu1 and not global}"
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "$TESTK" 2>&1 | tail -2
for i in 1 2 3; do
  for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_*.so; do
    echo -n "$(basename $lib) ${MODCOD:-QPSK-N_8/9}: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES="${SCAN_SIZES:-4096}" timeout 300 python tools/scan_batch.py ${MODCOD:-QPSK-N_8/9} ${IMPLEM:-NMS} 7 2>&1 | grep frames | tr '\n' ' '; echo
  done
done
