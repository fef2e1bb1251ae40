cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q --maxfail=15 2>&1 | tail -40 > $OUT/r06_g4_pytest.txt; tail -15 $OUT/r06_g4_pytest.txt
( time python bench.py > $OUT/r06_g4_bench.json 2> $OUT/r06_g4_bench.err ) 2>&1 | tail -4; tail -3 $OUT/r06_g4_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_g4_bench.json') if l.startswith('{')][-1])
print(d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['bound'], d['self_check'])
print(json.dumps(d['extra']['spa'])); print(json.dumps(d['extra']['ref_config'])); print(json.dumps(d['extra']['sync_located']))
for k in '23': print(k, {x: d['extra']['configs'][k].get(x) for x in ('ms','ldpc_kernel_ms','front_floor_ms','floor_ms','tail_over_floor')})
print(d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'))
PY
