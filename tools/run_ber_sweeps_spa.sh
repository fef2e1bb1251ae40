#!/bin/bash
# BER / FER sweeps with the reference's default decoder (--dec-implem SPA, 50 iterations, early stop) for every MODCOD, one MI355X; tables -> gpurun_out/berspa_*.txt / .json
set -u
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; python -m dvbs2_amd.sim "$@" --dec-implem SPA --dec-ite 50 --clones 3 -e 200 --json $OUT/berspa_$name.json 2>&1 | grep -v amdgpu.ids > $OUT/berspa_$name.txt; tail -3 $OUT/berspa_$name.txt; }
run qpsk_s_8_9   --mod-cod QPSK-S_8/9   -m 3.5 -M 4.21 -s 0.1 -F 8192 --max-frames 20000000
run qpsk_s_3_5   --mod-cod QPSK-S_3/5   -m 1.2 -M 1.81 -s 0.1 -F 8192 --max-frames 20000000
run 8psk_s_3_5   --mod-cod 8PSK-S_3/5   -m 2.6 -M 3.21 -s 0.1 -F 8192 --max-frames 20000000
run 8psk_s_8_9   --mod-cod 8PSK-S_8/9   -m 6.1 -M 6.81 -s 0.1 -F 8192 --max-frames 20000000
run 16apsk_s_8_9 --mod-cod 16APSK-S_8/9 -m 7.0 -M 7.71 -s 0.1 -F 8192 --max-frames 20000000 --est-type PERFECT
run 32apsk_s_3_4 --mod-cod 32APSK-S_3/4 -m 7.4 -M 8.41 -s 0.2 -F 8192 --max-frames 10000000 --est-type PERFECT
run qpsk_n_8_9   --mod-cod QPSK-N_8/9   -m 3.5 -M 3.91 -s 0.1 -F 4096 --max-frames 5000000
run 16apsk_n_8_9 --mod-cod 16APSK-N_8/9 -m 7.0 -M 7.41 -s 0.1 -F 4096 --max-frames 5000000 --est-type PERFECT
