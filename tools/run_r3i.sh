cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3i
for i in 1 2; do for e in "DVBS2HIP_SPA_MPITCH=1440" "DVBS2HIP_SPA_MPITCH=1440 DVBS2HIP_LDPC_SLOT_ALIGN=4" "DVBS2HIP_SPA_MPITCH=1536" "DVBS2HIP_SPA_MPITCH=1472"; do
  echo "== $e"; env $e python tools/bench_spa.py 16384 0 3 2>&1 | grep "SPA"
done; done > gpurun_out/r3i/pitch16k.txt 2>&1; cat gpurun_out/r3i/pitch16k.txt
python tools/bench_spa.py 16384 32768 3 2>&1 | grep -v amdgpu > gpurun_out/r3i/steady.txt; cat gpurun_out/r3i/steady.txt
