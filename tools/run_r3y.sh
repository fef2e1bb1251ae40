cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3y
AB_ROUNDS=3 AB_CMD='python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --self-check-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"ber\"][\"BE\"])"' bash tools/ab_variants.sh 2>&1 | tee gpurun_out/r3y/ab.txt
