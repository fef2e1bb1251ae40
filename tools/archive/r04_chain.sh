#!/bin/bash
# the fused chain's LDPC kernel against the bits socket's, same box; ablations of the chain's output phase (tools/bin/lib_abl*.so, wrong results by construction)
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_*.so 2>/dev/null | sort -V); do
 echo -n "$(basename $lib) socket: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
 echo -n "$(basename $lib) "; DVBS2HIP_LIB=$PWD/$lib timeout 300 python tools/chain_time.py 2>&1 | tail -1
done; done
