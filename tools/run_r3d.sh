cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3d
for e in "" "DVBS2HIP_LDPC_BLOCKS_PER_CU=1" "DVBS2HIP_LDPC_GRID_MAX=384" "DVBS2HIP_LDPC_GRID_MAX=320" "DVBS2HIP_LDPC_GRID_MAX=256" "DVBS2HIP_LDPC_GRID_MAX=448"; do
  echo "== $e"; env $e python tools/bench_spa.py 2>&1 | grep "N_8/9 4096 SPA"
done > gpurun_out/r3d/grid.txt 2>&1; cat gpurun_out/r3d/grid.txt
