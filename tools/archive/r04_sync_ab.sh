#!/bin/bash
# same-box A/B of the frame synchronizer: in-tree library against the variants under tools/bin; the sync parity tests of each first
cd "${GRAFT_REPO_ROOT:-.}"
for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_*.so 2>/dev/null | sort); do echo -n "parity $(basename $lib): "; DVBS2HIP_LIB=$PWD/$lib timeout 900 python -m pytest tests/test_sync_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -1; done
for i in 1 2 3; do for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_*.so 2>/dev/null | sort); do
 for c in "32APSK-S_3/4 4096" "QPSK-N_8/9 1024" "QPSK-S_8/9 4096"; do echo -n "$(basename $lib) "; DVBS2HIP_LIB=$PWD/$lib timeout 200 python tools/sync_time.py $c 2>&1 | tail -1; done
done; done
