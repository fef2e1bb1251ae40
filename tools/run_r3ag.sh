cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3ag
for m in static park; do echo "== $m"; DVBS2HIP_LDPC_FAST_MODE=$m DVBS2HIP_LIB=$PWD/tools/bin/lib_prof.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --self-check-steps 0 2>&1 | grep -A14 "phase prof" | tail -15; done | tee gpurun_out/r3ag/prof.txt
