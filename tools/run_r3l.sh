cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3l
( cd host && make -s ) > gpurun_out/r3l/hostmake.log 2>&1
AB_ROUNDS=2 AB_CMD='python tools/upfir_time.py; for m in 8PSK-N_8/9 QPSK-N_8/9 16APSK-N_8/9; do python tools/front_time.py $m; done; DVBS2HIP_FRONT_SINGLE=1 python tools/front_time.py QPSK-N_8/9' bash tools/ab_variants.sh > gpurun_out/r3l/ab.txt 2>&1; cat gpurun_out/r3l/ab.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3l/pytest.log 2>&1; tail -5 gpurun_out/r3l/pytest.log
