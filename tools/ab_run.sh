cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -3
AB_ROUNDS=3 AB_CMD='python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --self-check-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"ber\"][\"BE\"])"; python tools/bench_spa.py 0 16384 3 2>&1 | grep NMS' bash tools/ab_variants.sh 2>&1 | tee gpurun_out/ab/ab.txt
