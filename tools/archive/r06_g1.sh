cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -k "spa_tanh" 2>&1 | tail -15 > gpurun_out/r06_g1_pytest.txt; cat gpurun_out/r06_g1_pytest.txt
timeout 1500 tools/r06_spa_rules.sh 3000 2>&1 | tail -80
