cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3t
AB_GREP="SPA" AB_ROUNDS=2 AB_CMD="python tools/bench_spa.py 0 32768 3; python tools/spa_check.py 2>&1 | grep -c 'nan 0  hard diff 0'" bash tools/ab_variants.sh > gpurun_out/r3t/ab_short.txt 2>&1; cat gpurun_out/r3t/ab_short.txt
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "spa" > gpurun_out/r3t/pytest.log 2>&1; tail -3 gpurun_out/r3t/pytest.log
