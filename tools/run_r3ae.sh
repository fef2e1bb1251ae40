cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3ae
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r3ae/pytest.txt
AB_ROUNDS=3 AB_CMD='python bench.py --steps 20 --warmup 3 --no-cpu-baseline --self-check-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"ber\"][\"BE\"], d[\"extra\"][\"early_stop_fps\"], round(d[\"extra\"][\"fused_rx_chain\"][\"ms\"],3))"' bash tools/ab_variants.sh 2>&1 | tee gpurun_out/r3ae/ab.txt
for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_nl14.so; do echo "== $lib"; DVBS2HIP_LIB=$PWD/$lib python tools/bench_spa.py 16384 0 3 2>&1 | grep -v amdgpu; done | tee gpurun_out/r3ae/spa.txt
