cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3r
python tools/spa_check.py 2>&1 | grep -v amdgpu | cut -c1-104 > gpurun_out/r3r/spa_check.txt; cat gpurun_out/r3r/spa_check.txt
AB_GREP=SPA AB_ROUNDS=3 AB_CMD="python tools/bench_spa.py 4096 8192 3" bash tools/ab_variants.sh > gpurun_out/r3r/ab_msg4.txt 2>&1; cat gpurun_out/r3r/ab_msg4.txt
AB_GREP=SPA AB_ROUNDS=1 AB_CMD="python tools/bench_spa.py 16384 32768 3" bash tools/ab_variants.sh > gpurun_out/r3r/ab_msg4_steady.txt 2>&1; cat gpurun_out/r3r/ab_msg4_steady.txt
timeout 1200 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_refs_gpu.py -m gpu -x -q > gpurun_out/r3r/pytest.log 2>&1; tail -3 gpurun_out/r3r/pytest.log
