cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3k
AB_ROUNDS=2 AB_CMD='for m in 8PSK-N_8/9 8PSK-S_8/9 QPSK-N_8/9 QPSK-S_8/9; do python tools/front_time.py $m; done; DVBS2HIP_FRONT_SINGLE=1 python tools/front_time.py QPSK-N_8/9' bash tools/ab_variants.sh > gpurun_out/r3k/front_ab.txt 2>&1; cat gpurun_out/r3k/front_ab.txt
timeout 900 python -m pytest tests/test_front_gpu.py tests/test_unaligned_gpu.py tests/test_chain_gpu.py -m gpu -x -q > gpurun_out/r3k/pytest.log 2>&1; tail -3 gpurun_out/r3k/pytest.log
