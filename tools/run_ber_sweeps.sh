#!/bin/bash
# BER/FER sweeps on one MI355X (BASELINE configs 1, 3, 4-at-1-GPU, 5-modcod); tables -> gpurun_out/ber_*.txt
set -u
OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; python -m dvbs2_amd.sim "$@" --clones 1 --json $OUT/ber_$name.json 2>&1 | grep -v amdgpu.ids > $OUT/ber_$name.txt; tail -4 $OUT/ber_$name.txt; }
run qpsk_s_8_9_nms10      --mod-cod QPSK-S_8/9   -m 3.6 -M 4.41 -s 0.1 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 4000000
run qpsk_s_8_9_nms50_a875 --mod-cod QPSK-S_8/9   -m 3.6 -M 4.21 -s 0.1 --dec-implem NMS --dec-ite 50 --dec-alpha 0.875 -F 2048 --max-frames 4000000
run qpsk_s_3_5_nms10      --mod-cod QPSK-S_3/5   -m 1.4 -M 2.41 -s 0.2 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 1000000
run 8psk_s_3_5_nms10      --mod-cod 8PSK-S_3/5   -m 2.8 -M 4.01 -s 0.2 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 1000000
run 8psk_s_8_9_nms10      --mod-cod 8PSK-S_8/9   -m 6.4 -M 7.41 -s 0.2 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 1000000
run 16apsk_s_8_9_nms10    --mod-cod 16APSK-S_8/9 -m 7.2 -M 8.41 -s 0.2 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 1000000 --est-type PERFECT
run qpsk_n_8_9_nms10      --mod-cod QPSK-N_8/9   -m 3.4 -M 4.21 -s 0.1 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 1000000
run 16apsk_n_8_9_nms20    --mod-cod 16APSK-N_8/9 -m 7.0 -M 8.01 -s 0.2 --dec-implem NMS --dec-ite 20 -F 2048 --max-frames 500000 --est-type PERFECT
run 32apsk_s_3_4_nms10    --mod-cod 32APSK-S_3/4 -m 7.0 -M 9.01 -s 0.4 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 500000 --est-type PERFECT
