cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3o
python tools/spa_check.py 2>&1 | grep -v amdgpu | cut -c1-110 > gpurun_out/r3o/spa_check.txt; cat gpurun_out/r3o/spa_check.txt
for i in 1 2 3; do python tools/bench_spa.py 4096 8192 3 2>&1 | grep SPA; done > gpurun_out/r3o/spa4096.txt; cat gpurun_out/r3o/spa4096.txt
python tools/bench_spa.py 16384 32768 3 2>&1 | grep -v amdgpu > gpurun_out/r3o/spa_steady.txt; cat gpurun_out/r3o/spa_steady.txt
timeout 1200 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_refs_gpu.py tests/test_host_cpp.py -m gpu -x -q > gpurun_out/r3o/pytest.log 2>&1; tail -3 gpurun_out/r3o/pytest.log
