#!/bin/bash
# round 5: the reference's own configuration with 1 / 2 / 3 clones of the chain in flight (host/dvbs2_tx_rx_bb --clones), same box
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$REPO"
( cd host && make -s ) 2>&1 | tail -2
for rep in 1 2; do
for c in 1 2 3; do
  for mm in "3.6 3.61" "3.7 3.71" "3.8 3.81"; do
    set -- $mm
    echo -n "clones $c: "; ./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m $1 -M $2 -s 0.1 --dec-implem SPA --dec-ite 50 -F ${F:-8192} --clones $c 2>&1 | grep -E "^ +[0-9]"
  done
done
done
timeout 900 python -m pytest tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -5
