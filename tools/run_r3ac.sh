cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3ac
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3ac/pytest.txt
for i in 1 2; do for m in static park; do echo "== $m"; DVBS2HIP_LDPC_FAST_MODE=$m python tools/bench_spa.py 4096 0 3 2>&1 | grep -v amdgpu; done; done | tee gpurun_out/r3ac/spa_ab.txt
for m in static park; do echo "== $m 16384"; DVBS2HIP_LDPC_FAST_MODE=$m python tools/bench_spa.py 16384 0 3 2>&1 | grep -v amdgpu; done | tee -a gpurun_out/r3ac/spa_ab.txt
