cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3e
python tools/spa_check.py 2>&1 | grep -v amdgpu | cut -c1-100 > gpurun_out/r3e/spa_check.txt; cat gpurun_out/r3e/spa_check.txt
for i in 1 2; do for e in "DVBS2HIP_SPA_MPITCH=1536" "DVBS2HIP_SPA_MPITCH=1440" "DVBS2HIP_SPA_MPITCH=1472"; do
  echo "== $e"; env $e python tools/bench_spa.py 2>&1 | grep "SPA"
done; done > gpurun_out/r3e/pitch.txt 2>&1; cat gpurun_out/r3e/pitch.txt
