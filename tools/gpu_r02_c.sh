#!/bin/bash
# round-2 GPU pass C: GPU test-suite, per-kernel table, PMC passes of the non-LDPC kernels, the N = 64800 waterfall against ETSI's anchor.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/c_hostmake.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $OUT/c_pytest.log 2>&1; echo "pytest rc $?"; tail -6 $OUT/c_pytest.log
python tools/bench_kernels.py $OUT/c_kernels.json > $OUT/c_kernels.log 2>&1; tail -2 $OUT/c_kernels.log
bash tools/profile_kernels.sh > $OUT/c_profile_kernels.log 2>&1; tail -2 $OUT/c_profile_kernels.log
python -m dvbs2_amd.sim --mod-cod QPSK-N_8/9 -m 3.40 -M 3.86 -s 0.05 --dec-implem SPA --dec-ite 50 -F 4096 --max-frames 600000 --json $OUT/c_waterfall_qpsk_n_spa50.json > $OUT/c_waterfall_qpsk_n_spa50.txt 2>&1; tail -12 $OUT/c_waterfall_qpsk_n_spa50.txt
python -m dvbs2_amd.sim --mod-cod QPSK-N_8/9 -m 3.40 -M 4.21 -s 0.1 --dec-implem NMS --dec-ite 10 -F 4096 --max-frames 400000 --json $OUT/c_waterfall_qpsk_n_nms10.json > $OUT/c_waterfall_qpsk_n_nms10.txt 2>&1; tail -10 $OUT/c_waterfall_qpsk_n_nms10.txt
