cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -8 | tee $OUT/r06_g15_pytest.txt
