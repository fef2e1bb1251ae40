#!/bin/bash
# same-box A/B of the front end: in-tree library against the variants under tools/bin (DVBS2HIP_LIB); parity tests of the in-tree one first
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_front_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in ${FRONT_AB_TEST_LIBS:-}; do echo "parity with $v:"; DVBS2HIP_LIB=$PWD/tools/bin/lib_$v.so timeout 900 python -m pytest tests/test_front_gpu.py -m gpu -x -q 2>&1 | tail -1; done
for i in 1 2 3; do for lib in dvbs2_amd/lib/libdvbs2hip.so $(ls tools/bin/lib_*.so | sort); do
 for mc in ${FRONT_AB_MODCODS:-16APSK-N_8/9 32APSK-S_3/4 16APSK-S_8/9 8PSK-N_8/9 8PSK-S_8/9}; do echo -n "$(basename $lib) "; DVBS2HIP_LIB=$PWD/$lib timeout 120 python tools/front_time.py $mc 2>&1 | tail -1; done
done; done
