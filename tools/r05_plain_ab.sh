#!/bin/bash
# round 5: the bits-socket form (the bench line's step) of the in-tree library against tools/bin/lib_*.so variants, same box, alternating
cd "${GRAFT_REPO_ROOT:-.}"
for i in $(seq 1 ${AB_ROUNDS:-4}); do
  for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_*.so | grep -v lib_po | sort -V); do
    echo -n "$(basename $lib) N: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 9 2>&1 | grep frames | tr '\n' ' '; echo
    if [ "${AB_SHORT:-1}" = 1 ]; then echo -n "$(basename $lib) S: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=16384 timeout 300 python tools/scan_batch.py QPSK-S_8/9 NMS 9 2>&1 | grep frames | tr '\n' ' '; echo; fi
  done
done
