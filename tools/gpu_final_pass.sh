#!/bin/bash
# The measurement passes behind a round's committed numbers, on the GPU box (gpurun -- 'bash tools/gpu_final_pass.sh A' then, after summarizing, '... B'):
#   A  bench line + per-kernel table + rocprofv3 passes of bench.py and of the LDPC instantiations + sum-product rates + the reference's own configuration
#      afterwards, in the build container: cp gpurun_out/kernels.json results/<tag>/; python tools/summarize_profiles.py <tag>; python tools/summarize_ldpc_variants.py <tag>;
#      python tools/summarize_ref_config.py <tag> (rows of results/<tag>/ref_config_spa50.md)
#   B  the bench line again (its roofline.traffic needs A's stamp in profiles/ldpc_pmc_traffic.json), the 1-rank RCCL logs, the whole -m gpu suite
#      afterwards: cp gpurun_out/bench.json profiles/<tag>_bench.json; cp gpurun_out/*_1rank.log profiles/; python tools/make_design_tables.py <tag> --write
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/final
case "${1:-A}" in
A) bash tools/gpu_round.sh bench kernels prof profk > gpurun_out/final/round_a.log 2>&1; tail -3 gpurun_out/final/round_a.log
   bash tools/profile_ldpc_variants.sh > gpurun_out/final/profile_lv.log 2>&1; tail -3 gpurun_out/final/profile_lv.log
   python tools/bench_spa.py 4096 8192 3 2>&1 | grep -v amdgpu > gpurun_out/final/bench_spa_4096.txt; python tools/bench_spa.py 16384 32768 3 2>&1 | grep -v amdgpu > gpurun_out/final/bench_spa_steady.txt
   cat gpurun_out/final/bench_spa_4096.txt gpurun_out/final/bench_spa_steady.txt
   bash tools/ref_config_spa50.sh > gpurun_out/final/ref_config.txt 2>&1; grep -E "^ +[0-9]" gpurun_out/final/ref_config.txt | head -3 ;;
B) bash tools/gpu_round.sh bench rccl tests > gpurun_out/final/round_b.log 2>&1; tail -8 gpurun_out/final/round_b.log ;;
esac
