cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3ad
for m in static park; do echo "== $m"; DVBS2HIP_LDPC_FAST_MODE=$m python tools/det_check.py 4096 0.42 QPSK-N_8/9 2>&1 | grep -v amdgpu; done | tee gpurun_out/r3ad/det.txt
echo "== default, short"; python tools/det_check.py 8192 0.42 QPSK-S_8/9 2>&1 | grep -v amdgpu | tee -a gpurun_out/r3ad/det.txt
