cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3u
AB_GREP="N_8/9 16384 SPA" AB_ROUNDS=2 AB_CMD="python tools/bench_spa.py 16384 0 2" bash tools/ab_variants.sh > gpurun_out/r3u/ab_st.txt 2>&1; cat gpurun_out/r3u/ab_st.txt
