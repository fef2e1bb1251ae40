#!/bin/bash
# round-2 GPU pass E: GPU test-suite, per-kernel table with the locked-stream sync benchmark, PMC passes of the non-LDPC kernels
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/e_hostmake.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $OUT/e_pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/e_pytest.log
python tools/bench_kernels.py $OUT/e_kernels.json > $OUT/e_kernels.log 2>&1; tail -2 $OUT/e_kernels.log
bash tools/profile_kernels.sh > $OUT/e_profile_kernels.log 2>&1
