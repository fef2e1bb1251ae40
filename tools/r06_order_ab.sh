cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_ldpc_gpu.py tests/test_host_cpp.py -q -k "work_queue or sweep_order or cpp" 2>&1 | tail -8 | tee $OUT/r06_g6_pytest.txt
for ord in 0 1; do for c in 1 3; do
  echo "order=$ord clones=$c: $(DVBS2HIP_LDPC_ORDER=$ord ./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.6 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F 8192 --clones $c -e 100000000 --max-frames 2000000 | grep -E '^ +[0-9]' | tr '\n' ' ')"
done; done 2>&1 | tee $OUT/r06_g6_order.txt
for ord in 0 1; do echo "order=$ord N: $(DVBS2HIP_LDPC_ORDER=$ord ./host/dvbs2_tx_rx_bb --mod-cod QPSK-N_8/9 -m 3.9 -M 4.11 -s 0.1 --dec-implem SPA --dec-ite 50 -F 4096 --clones 1 -e 100000000 --max-frames 300000 | grep -E '^ +[0-9]' | tr '\n' ' ')"; done 2>&1 | tee -a $OUT/r06_g6_order.txt
for ord in 0 1; do echo "order=$ord NMS-N es: $(DVBS2HIP_LDPC_ORDER=$ord ./host/dvbs2_tx_rx_bb --mod-cod QPSK-N_8/9 -m 3.9 -M 4.11 -s 0.1 --dec-implem NMS --dec-ite 10 -F 4096 --clones 1 -e 100000000 --max-frames 600000 | grep -E '^ +[0-9]' | tr '\n' ' ')"; done 2>&1 | tee -a $OUT/r06_g6_order.txt
python bench.py --no-cpu-baseline --no-live-pmc > $OUT/r06_g6_bench.json 2> $OUT/r06_g6_bench.err; tail -2 $OUT/r06_g6_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_g6_bench.json') if l.startswith('{')][-1])
print(d['value']/1e9, d['ms_per_step'])
for r in d['extra']['configs']['4']['per_F']: print({k: r.get(k) for k in ('frames','latency_ms_median','latency_ms_graph','latency_ms_early_stop','latency_ms_early_stop_graph','frames_decoded_exactly','frames_decoded_exactly_graph')})
print(json.dumps(d['extra']['ref_config'])); print(d['extra']['early_stop_fps'])
PY
