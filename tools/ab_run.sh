cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -2
AB_ROUNDS=3 AB_CMD='python tools/bench_spa.py 0 16384 3 2>&1 | grep " NMS"; python tools/bench_32apsk.py 2>&1 | grep NMS' bash tools/ab_variants.sh 2>&1 | tee gpurun_out/ab/ab.txt
