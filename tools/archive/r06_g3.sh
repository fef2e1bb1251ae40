cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/r06_g3_pytest.txt
python tools/bench_spa.py 4096 8192 3 2>&1 | grep -v amdgpu | tee $OUT/r06_g3_bench_spa.txt
