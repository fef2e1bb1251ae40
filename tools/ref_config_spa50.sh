#!/bin/bash
# The reference's own configuration (the command lines of refs/TX_RX_BB/*.txt: SPA, 50 iterations, early stop, 100 frame errors) through the C++ simulator on
# one MI355X: the refs-format tables -> gpurun_out/ref_config_*.txt (GPU box, after `make -C host`)
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
run() { name=$1; shift; ./host/dvbs2_tx_rx_bb "$@" --dec-implem SPA --dec-ite 50 -F 8192 > $OUT/ref_config_$name.txt 2>&1; grep -E "^ +[0-9]" $OUT/ref_config_$name.txt; }
run qpsk_8_9   --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1
run qpsk_3_5   --mod-cod QPSK-S_3/5   -m 1.3 -M 1.51 -s 0.1
run 8psk_3_5   --mod-cod 8PSK-S_3/5   -m 2.7 -M 3.01 -s 0.1
run 8psk_8_9   --mod-cod 8PSK-S_8/9   -m 6.2 -M 6.51 -s 0.1
run 16apsk_8_9 --mod-cod 16APSK-S_8/9 -m 7.1 -M 7.51 -s 0.1 --est-type PERFECT
# one clone of the chain (the loop of rounds 2-4: the same batches seed for seed) and four, on the rows the rounds have tracked
run c1_qpsk_8_9 --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1 --clones 1
run c2_qpsk_8_9 --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1 --clones 2
run c4_qpsk_8_9 --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1 --clones 4
