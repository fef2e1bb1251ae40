cd "${GRAFT_REPO_ROOT:-.}"
echo "== 1 WG per CU"; DVBS2HIP_LDPC_BLOCKS_PER_CU=1 DET_ITE=2 python tools/det_check.py 4096 0.50 QPSK-S_8/9 2>&1 | grep -v amdgpu | grep SPA | cut -c1-160
echo "== 2 WG per CU, 512 frames (one frame per WG)"; DET_ITE=2 python tools/det_check.py 512 0.50 QPSK-S_8/9 2>&1 | grep -v amdgpu | grep SPA | cut -c1-160
echo "== 2 WG per CU, 256 frames"; DET_ITE=2 python tools/det_check.py 256 0.50 QPSK-S_8/9 2>&1 | grep -v amdgpu | grep SPA | cut -c1-160
echo "== grid max 256 (DVBS2HIP_LDPC_GRID_MAX), 4096 frames"; DVBS2HIP_LDPC_GRID_MAX=256 DET_ITE=2 python tools/det_check.py 4096 0.50 QPSK-S_8/9 2>&1 | grep -v amdgpu | grep SPA | cut -c1-160
