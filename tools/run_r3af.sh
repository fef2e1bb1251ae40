cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3af
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3af/pytest.txt
B='python bench.py --steps 20 --warmup 3 --no-cpu-baseline --self-check-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"ber\"][\"BE\"], d[\"roofline\"][\"kernel\"], d[\"extra\"][\"early_stop_fps\"], round(d[\"extra\"][\"fused_rx_chain\"][\"ms\"],3))"'
for i in 1 2 3; do for m in static park4 park; do echo -n "$m "; DVBS2HIP_LDPC_FAST_MODE=$m bash -c "$B"; done; done 2>&1 | tee gpurun_out/r3af/ab.txt
