#!/bin/bash
# Round 4: mode 6 (one frame per CU, k_ldpc_cu1.hip) -- parity against the oracle, timing against mode 5 on the same box, per-phase cycle profile.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "image_modes and NMS and (cu1 or park-5)" > $OUT/r04_cu1_test.log 2>&1; tail -15 $OUT/r04_cu1_test.log
for i in 1 2; do
  for m in park cu1; do
    echo "== $m"; DVBS2HIP_LDPC_FAST_MODE=$m SCAN_SIZES="${SCAN_SIZES:-256 4096 8192}" timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 5 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/r04_cu1_time.txt 2>&1
cat $OUT/r04_cu1_time.txt
DVBS2HIP_LIB=$PWD/tools/bin/lib_phasec.so DVBS2HIP_LDPC_FAST_MODE=cu1 SCAN_SIZES="4096" timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 1 > $OUT/r04_cu1_phase.txt 2>&1; tail -14 $OUT/r04_cu1_phase.txt
