#!/bin/bash
# long runs at the bottom of the waterfall: tens of millions of frames per point, looking for an error floor (none expected at these rates)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; python -m dvbs2_amd.sim "$@" --json $OUT/floor_$name.json 2>&1 | grep -v amdgpu.ids > $OUT/floor_$name.txt; tail -3 $OUT/floor_$name.txt; }
run qpsk_n_8_9_nms10   --mod-cod QPSK-N_8/9   -m 4.10 -M 4.31 -s 0.1 --dec-implem NMS --dec-ite 10 -F 4096 --max-frames 20000000
run qpsk_n_8_9_spa50   --mod-cod QPSK-N_8/9   -m 3.80 -M 3.91 -s 0.1 --dec-implem SPA --dec-ite 50 -F 4096 --max-frames 10000000
run qpsk_s_8_9_nms10   --mod-cod QPSK-S_8/9   -m 4.40 -M 4.61 -s 0.1 --dec-implem NMS --dec-ite 10 -F 8192 --max-frames 50000000
run 16apsk_n_8_9_nms20 --mod-cod 16APSK-N_8/9 -m 8.00 -M 8.21 -s 0.2 --dec-implem NMS --dec-ite 20 -F 4096 --max-frames 10000000 --est-type PERFECT
run 32apsk_s_3_4_nms10 --mod-cod 32APSK-S_3/4 -m 9.00 -M 9.21 -s 0.2 --dec-implem NMS --dec-ite 10 -F 8192 --max-frames 30000000 --est-type PERFECT
