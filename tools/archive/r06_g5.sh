cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -q -k "sweep_order" 2>&1 | tail -12 | tee $OUT/r06_g5_pytest.txt
timeout 3000 tools/r06_natural.sh 3000 2>&1 | tail -100
