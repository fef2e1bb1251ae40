cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3aj
bash tools/gpu_round.sh bench kernels prof > gpurun_out/r3aj/round.log 2>&1; tail -3 gpurun_out/r3aj/round.log
bash tools/profile_ldpc_variants.sh > gpurun_out/r3aj/profile_lv.log 2>&1; tail -3 gpurun_out/r3aj/profile_lv.log
python tools/bench_spa.py 4096 8192 3 2>&1 | grep -v amdgpu > gpurun_out/r3aj/bench_spa_4096.txt; python tools/bench_spa.py 16384 32768 3 2>&1 | grep -v amdgpu > gpurun_out/r3aj/bench_spa_steady.txt; cat gpurun_out/r3aj/bench_spa_4096.txt gpurun_out/r3aj/bench_spa_steady.txt
bash tools/ref_config_spa50.sh > gpurun_out/r3aj/ref_config.txt 2>&1; grep -E "^ +[0-9]" gpurun_out/r3aj/ref_config.txt | head -3
