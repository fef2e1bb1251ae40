cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3v
AB_GREP="SPA" AB_ROUNDS=2 AB_CMD="python tools/bench_spa.py 4096 32768 3" bash tools/ab_variants.sh > gpurun_out/r3v/ab_st2.txt 2>&1; cat gpurun_out/r3v/ab_st2.txt
