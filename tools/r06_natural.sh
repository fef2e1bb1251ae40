#!/bin/bash
# Round 6: the reference's decoder as recalled -- tanh-product check node, natural row order -- and `--dec-implem SPA` (exact node + AFF3CT's cap) in both sweep orders, the five
# command lines of refs/TX_RX_BB/*.txt to $FE frame errors per row -> gpurun_out/r06_{clip,tanhnat,clipnat}_<trace>.txt (tools/refs_pooled.py, tools/make_spa_rules_md.py);
# then QC layers against natural order on the SAME frames (one Eb/N0 per invocation, fixed frame counts) -> gpurun_out/r06_paired_sched.txt
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
FE=${1:-3000}
( cd host && make -s ) 2>&1 | tail -2
run() { tag=$1; name=$2; shift 2; timeout 900 ./host/dvbs2_tx_rx_bb "$@" --dec-ite 50 -e $FE --max-frames 40000000 > $OUT/r06_${tag}_$name.txt 2>&1; }
traces() { tag=$1; shift
  run $tag qpsk_8_9   --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1 "$@"
  run $tag qpsk_3_5   --mod-cod QPSK-S_3/5   -m 1.3 -M 1.51 -s 0.1 "$@"
  run $tag 8psk_3_5   --mod-cod 8PSK-S_3/5   -m 2.7 -M 3.01 -s 0.1 "$@"
  run $tag 8psk_8_9   --mod-cod 8PSK-S_8/9   -m 6.2 -M 6.51 -s 0.1 "$@"
  run $tag 16apsk_8_9 --mod-cod 16APSK-S_8/9 -m 7.1 -M 7.51 -s 0.1 --est-type PERFECT "$@"
  echo "== $tag"; grep -hE "^ +[0-9]" $OUT/r06_${tag}_*.txt; }
traces clip    --dec-implem SPA -F 8192
traces tanhnat --dec-implem SPA_TANH --dec-sched NATURAL -F 32768 --clones 2
traces clipnat --dec-implem SPA --dec-sched NATURAL -F 32768 --clones 2
: > $OUT/r06_paired_sched.txt
paired() { modcod=$1; eb=$2; frames=$3; shift 3
  for v in "SPA QC" "SPA NATURAL" "SPA_TANH NATURAL"; do
    set -- $v
    l=$(timeout 600 ./host/dvbs2_tx_rx_bb --mod-cod $modcod -m $eb -M $(python3 -c "print($eb + 0.01)") -s 0.1 --dec-implem $1 --dec-sched $2 --dec-ite 50 -F 32768 --clones 2 -e 100000000 --max-frames $frames $EST | grep -E "^ +[0-9]")
    echo "$modcod $eb $1 $2 $l" >> $OUT/r06_paired_sched.txt
  done; }
EST=""
paired QPSK-S_8/9 3.7 196608; paired QPSK-S_8/9 3.8 1572864
paired QPSK-S_3/5 1.4 196608; paired QPSK-S_3/5 1.5 2162688
paired 8PSK-S_3/5 2.9 294912; paired 8PSK-S_3/5 3.0 1966080
paired 8PSK-S_8/9 6.3 131072; paired 8PSK-S_8/9 6.5 1769472
EST="--est-type PERFECT"
paired 16APSK-S_8/9 7.3 196608; paired 16APSK-S_8/9 7.5 2949120
cat $OUT/r06_paired_sched.txt
