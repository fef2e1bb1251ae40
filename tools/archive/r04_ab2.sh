cd "${GRAFT_REPO_ROOT:-.}"
DVBS2HIP_LIB=$PWD/tools/bin/lib_athyb.so timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "image_modes and NMS and not cu1 and not global" 2>&1 | tail -2
for i in 1 2 3; do for m in "" park4 static; do for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_athyb.so; do
 echo -n "$(basename $lib) mode=${m:-default}: "; DVBS2HIP_LDPC_FAST_MODE=$m DVBS2HIP_LIB=$PWD/$lib SCAN_SIZES=4096 timeout 300 python tools/scan_batch.py QPSK-N_8/9 NMS 7 2>&1 | grep frames | tr '\n' ' '; echo
done; done; done
