#!/bin/bash
# same-box A/B: in-tree library against the variants under tools/bin -- normal frames (modes park / static), short frames (three codes); parity tests first
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "not SPA and not cu1" 2>&1 | tail -2
for i in 1 2 3; do for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_*.so; do
 for m in park static; do echo -n "$(basename $lib) N $m: "; DVBS2HIP_LDPC_FAST_MODE=$m DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo; done
 for mc in QPSK-S_8/9 32APSK-S_3/4 QPSK-S_3/5; do echo -n "$(basename $lib) $mc: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=16384 timeout 300 python tools/scan_batch.py $mc NMS 7 2>&1 | grep frames | tr '\n' ' '; echo; done
done; done
