cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
for c in 1 2 3; do echo "clones=$c: $(./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.6 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F 8192 --clones $c -e 100000000 --max-frames 2000000 | grep -E '^ +[0-9]' | awk -F'\\|\\|' '{print $3}' | tr '\n' ' ')"; done 2>&1 | tee $OUT/r06_g13_refcfg.txt
./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.8 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F 8192 --clones 1 -e 100000000 --max-frames 1589248 | grep -E '^ +[0-9]' | tee -a $OUT/r06_g13_refcfg.txt
