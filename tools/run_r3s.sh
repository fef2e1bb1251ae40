cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3s
bash tools/gpu_round.sh prof > gpurun_out/r3s/prof.log 2>&1; tail -2 gpurun_out/r3s/prof.log
bash tools/profile_ldpc_variants.sh > gpurun_out/r3s/profile_lv.log 2>&1; tail -3 gpurun_out/r3s/profile_lv.log
bash tools/gpu_round.sh refs > gpurun_out/r3s/refs.log 2>&1; tail -6 gpurun_out/r3s/refs.log
