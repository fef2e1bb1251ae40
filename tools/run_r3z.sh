cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3z
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3z/pytest.txt; cat gpurun_out/r3z/pytest.txt
B='python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --self-check-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"ber\"], d[\"roofline\"][\"kernel\"])"'
for i in 1 2 3; do
  echo -n "mode3 "; DVBS2HIP_LDPC_FAST_MODE=static bash -c "$B"
  echo -n "mode4 "; bash -c "$B"
done 2>&1 | tee gpurun_out/r3z/ab.txt
