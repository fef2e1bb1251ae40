cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3j
( cd host && make -s ) > gpurun_out/r3j/hostmake.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r3j/pytest.log 2>&1; tail -5 gpurun_out/r3j/pytest.log
bash tools/profile_ldpc_variants.sh > gpurun_out/r3j/profile_lv.log 2>&1; tail -8 gpurun_out/r3j/profile_lv.log
