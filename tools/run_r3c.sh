cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3c
AB_GREP="N_8/9 4096 SPA" AB_ROUNDS=1 bash tools/ab_variants.sh > gpurun_out/r3c/abl_spa.txt 2>&1; cat gpurun_out/r3c/abl_spa.txt
