cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3a
python tools/spa_check.py > gpurun_out/r3a/spa_check.txt 2>&1; tail -12 gpurun_out/r3a/spa_check.txt
python tools/bench_spa.py > gpurun_out/r3a/bench_spa.txt 2>&1; cat gpurun_out/r3a/bench_spa.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py tests/test_refs_gpu.py -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; tail -5 gpurun_out/r3a/pytest.log
