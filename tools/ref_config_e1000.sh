#!/bin/bash
# The reference's own configuration (tools/ref_config_spa50.sh) to 1000 frame errors per row instead of the reference's 100: FER / BER of the GPU chain to +-3 % (one sigma)
# beside the reference's traces -> gpurun_out/ref1000_*.txt; REF_CONFIG_PREFIX=ref1000_ python tools/summarize_ref_config.py r05 prints the rows
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
run() { name=$1; shift; ./host/dvbs2_tx_rx_bb "$@" --dec-implem SPA --dec-ite 50 -F 8192 -e 1000 --max-frames 20000000 > $OUT/ref1000_$name.txt 2>&1; grep -E "^ +[0-9]" $OUT/ref1000_$name.txt; }
run qpsk_8_9   --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1
run qpsk_3_5   --mod-cod QPSK-S_3/5   -m 1.3 -M 1.51 -s 0.1
run 8psk_3_5   --mod-cod 8PSK-S_3/5   -m 2.7 -M 3.01 -s 0.1
run 8psk_8_9   --mod-cod 8PSK-S_8/9   -m 6.2 -M 6.51 -s 0.1
run 16apsk_8_9 --mod-cod 16APSK-S_8/9 -m 7.1 -M 7.51 -s 0.1 --est-type PERFECT
