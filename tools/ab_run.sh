cd "${GRAFT_REPO_ROOT:-.}"
for lib in tools/bin/lib_w1.so tools/bin/lib_w2.so tools/bin/lib_w4.so; do echo "== $lib"; for rep in 1 2; do DVBS2HIP_LIB=$PWD/$lib DET_ITE=3 python tools/det_check.py 4096 0.50 QPSK-S_8/9 2>&1 | grep -v amdgpu | grep SPA | cut -c30-150; done; done
