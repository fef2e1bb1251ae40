cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3q
python tools/spa_check.py 2>&1 | grep -v amdgpu | cut -c1-104 > gpurun_out/r3q/spa_check.txt; cat gpurun_out/r3q/spa_check.txt
for i in 1 2; do python tools/bench_spa.py 4096 8192 3 2>&1 | grep SPA; done > gpurun_out/r3q/spa4096.txt; cat gpurun_out/r3q/spa4096.txt
bash tools/ref_config_spa50.sh > gpurun_out/r3q/ref_config.txt 2>&1; cat gpurun_out/r3q/ref_config.txt
timeout 1200 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_refs_gpu.py -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; tail -3 gpurun_out/r3q/pytest.log
