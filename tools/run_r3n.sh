cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3n
bash tools/gpu_round.sh prof bench rccl > gpurun_out/r3n/round.log 2>&1; tail -12 gpurun_out/r3n/round.log
