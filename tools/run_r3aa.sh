cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3aa
bash tools/gpu_round.sh tests 2>&1 | tail -6 | tee gpurun_out/r3aa/tests.txt
cp gpurun_out/tests_pytest.log gpurun_out/r3aa/ 2>/dev/null
