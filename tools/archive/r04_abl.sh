#!/bin/bash
# ablations of the min-sum layer (-DW8_ABL=..., wrong results by construction: timing only), same box, against the in-tree library
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_abl*.so | sort -V); do
 echo -n "$(basename $lib) N: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
 echo -n "$(basename $lib) S: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=16384 timeout 300 python tools/scan_batch.py QPSK-S_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
done; done
