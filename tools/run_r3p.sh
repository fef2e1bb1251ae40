cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3p
timeout 1200 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_refs_gpu.py tests/test_host_cpp.py -m gpu -x -q --durations=12 > gpurun_out/r3p/pytest.log 2>&1; tail -22 gpurun_out/r3p/pytest.log
