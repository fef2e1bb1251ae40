cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3g
for i in 1 2; do for e in "DVBS2HIP_SPA_MPITCH=1536" "DVBS2HIP_SPA_MPITCH=1440" "DVBS2HIP_SPA_MPITCH=1472" "DVBS2HIP_SPA_MPITCH=1440 DVBS2HIP_LDPC_SLOT_ALIGN=128" "DVBS2HIP_SPA_MPITCH=1440 DVBS2HIP_LDPC_SLOT_ALIGN=4"; do
  echo "== $e"; env $e python tools/bench_spa.py 2>&1 | grep "SPA"
done; done > gpurun_out/r3g/pitch.txt 2>&1; cat gpurun_out/r3g/pitch.txt
