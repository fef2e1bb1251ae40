cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3m
python tools/bench_kernels.py gpurun_out/r3m/kernels.json > gpurun_out/r3m/kernels.log 2>&1; tail -3 gpurun_out/r3m/kernels.log
bash tools/profile_kernels.sh > gpurun_out/r3m/profile_kernels.log 2>&1; tail -3 gpurun_out/r3m/profile_kernels.log
