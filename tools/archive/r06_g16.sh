cd "${GRAFT_REPO_ROOT:-.}"
( cd host && make -s ) 2>&1 | tail -2
for v in "SPA QC 8192" "SPA_TANH QC 8192" "SPA NATURAL 32768" "SPA_EXACT QC 8192"; do set -- $v
  echo "$1 $2: $(timeout 900 ./host/dvbs2_tx_rx_bb --mod-cod 8PSK-S_3/5 -m 3.2 -M 3.21 -s 0.1 --dec-implem $1 --dec-sched $2 --dec-ite 50 -F $3 --clones 2 -e 100000000 --max-frames 4000000 | grep -E '^ +[0-9]')"
done 2>&1 | tee gpurun_out/r06_8psk35_32.txt
