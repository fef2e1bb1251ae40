cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q --maxfail=20 2>&1 | tail -25 > $OUT/r06_g7_pytest.txt; tail -12 $OUT/r06_g7_pytest.txt
