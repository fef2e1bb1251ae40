#!/bin/bash
# round-2 GPU pass B: whole GPU test-suite, bench line, per-kernel table, rocprofv3 passes of bench.py (LDPC) and of the other kernels.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/b_hostmake.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $OUT/b_pytest.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/b_pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/b_bench.json 2> $OUT/b_bench.err; tail -c 600 $OUT/b_bench.json
python tools/bench_kernels.py $OUT/b_kernels.json > $OUT/b_kernels.log 2>&1; tail -3 $OUT/b_kernels.log
bash tools/profile_gpu.sh > $OUT/b_profile.log 2>&1
bash tools/profile_kernels.sh > $OUT/b_profile_kernels.log 2>&1; tail -3 $OUT/b_profile_kernels.log
