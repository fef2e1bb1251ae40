#!/bin/bash
# GPU box: the ablation builds (tools/build_ablations.sh) against the in-tree library, alternating, bits socket, QPSK-N_8/9 4096 frames (and QPSK-S_8/9 16384);
# then: python tools/summarize_ablations.py gpurun_out/<file> -> profiles/ldpc_ablation.json (stamped with the kernel sources' hash; bench.py's roofline reads it)
cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2 3; do for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_abl*.so | sort -V); do
 echo -n "$(basename $lib) N: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
 if [ $i = 1 ]; then echo -n "$(basename $lib) S: "; DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=16384 timeout 300 python tools/scan_batch.py QPSK-S_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo; fi
done; done
