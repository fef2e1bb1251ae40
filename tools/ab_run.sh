cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
B='python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --self-check-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"ms_per_step\"],3), d[\"ber\"][\"BE\"])"'
for i in 1 2 3; do for o in 0 1 2 5 10 6 9; do echo -n "order $o: "; DVBS2HIP_LDPC_HYB_ORDER=$o bash -c "$B"; done; done 2>&1 | tee gpurun_out/ab/order.txt
