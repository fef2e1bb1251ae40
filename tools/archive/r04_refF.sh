cd "${GRAFT_REPO_ROOT:-.}"; (cd host && make -s) > /dev/null 2>&1
for F in 8192 16384 32768; do echo "-F $F"; ./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.6 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F $F 2>&1 | grep -E "^ +[0-9]"; done
