cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3ab
for m in static park; do
  DVBS2HIP_LDPC_FAST_MODE=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --self-check-steps 0 2>/dev/null | tail -1 > gpurun_out/r3ab/bench_$m.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r3ab/bench_$m.json"))
print("$m", round(d["ms_per_step"],3), d["extra"]["early_stop_fps"], d["extra"]["fused_rx_chain"]["ms"], d["extra"]["hard_batch_fixed_10_ite"]["ms"])
PY
done
python tools/bench_spa.py 4096 0 3 2>&1 | grep -v amdgpu
