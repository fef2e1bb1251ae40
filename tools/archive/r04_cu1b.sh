#!/bin/bash
# Round 4: variants of mode 6 against each other and mode 5 on one box (libraries under tools/bin/, DVBS2HIP_LIB), then the per-phase profile
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "image_modes and NMS and cu1" > $OUT/r04_cu1_test.log 2>&1; tail -3 $OUT/r04_cu1_test.log
for lib in tools/bin/lib_ha*.so; do
  echo "== test $lib"; DVBS2HIP_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "image_modes and NMS and cu1" 2>&1 | tail -1
done
for i in 1 2; do
  echo "== park"; DVBS2HIP_LDPC_FAST_MODE=park SCAN_SIZES="4096" timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 5 2>&1 | grep frames
  for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_ha*.so; do
    echo "== cu1 $lib"; DVBS2HIP_LIB=$PWD/$lib DVBS2HIP_LDPC_FAST_MODE=cu1 SCAN_SIZES="4096" timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 5 2>&1 | grep frames
  done
done > $OUT/r04_cu1_time.txt 2>&1
cat $OUT/r04_cu1_time.txt
for lib in tools/bin/lib_phase*.so; do
DVBS2HIP_LIB=$PWD/$lib DVBS2HIP_LDPC_FAST_MODE=cu1 SCAN_SIZES="4096" timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 1 > $OUT/r04_cu1_$(basename $lib .so).txt 2>&1; tail -17 $OUT/r04_cu1_$(basename $lib .so).txt
done
