cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3b
python tools/spa_check.py > gpurun_out/r3b/spa_check.txt 2>&1; grep -v amdgpu gpurun_out/r3b/spa_check.txt | cut -c1-120
AB_GREP=SPA AB_ROUNDS=2 bash tools/ab_variants.sh > gpurun_out/r3b/ab_spa.txt 2>&1; cat gpurun_out/r3b/ab_spa.txt
timeout 900 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "spa or golden or plain" > gpurun_out/r3b/pytest.log 2>&1; tail -3 gpurun_out/r3b/pytest.log
