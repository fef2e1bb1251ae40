cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3ak
bash tools/gpu_round.sh bench rccl tests > gpurun_out/r3ak/round.log 2>&1; tail -8 gpurun_out/r3ak/round.log
