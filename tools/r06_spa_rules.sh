#!/bin/bash
# Round 6, VERDICT r5 item 1: the two sum-product check-node rules against the reference's traces.
#   (a) the five command lines of refs/TX_RX_BB/*.txt through host/dvbs2_tx_rx_bb with --dec-implem SPA_EXACT (rounds 1-5's rule) and SPA_TANH (AFF3CT's saturating form), -e $FE frame errors per row
#       -> gpurun_out/r06_${tag}_<trace>.txt, read by tools/refs_pooled.py
#   (b) PAIRED: every row once more with a fixed number of frames and the same seeds under both rules (one Eb/N0 per invocation: the batch counter -- the noise seed -- then
#       starts at 0 for both) -> gpurun_out/r06_paired.txt: frames, FE exact, FE tanh
# usage: tools/r06_spa_rules.sh [FE=3000] [extra args for the simulator, e.g. --noise-gen FAST]      TAGSUFFIX=_fast to keep the files apart
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
FE=${1:-3000}; shift
EXTRA="$@"
SFX=${TAGSUFFIX:-}
( cd host && make -s ) 2>&1 | tail -2
run() { tag=$1; implem=$2; name=$3; shift 3; ./host/dvbs2_tx_rx_bb "$@" --dec-implem $implem --dec-ite 50 -F 8192 -e $FE --max-frames 40000000 $EXTRA > $OUT/r06_${tag}${SFX}_$name.txt 2>&1; }
for v in "exact SPA_EXACT" "tanh SPA_TANH"; do      # (the capped default, `SPA`, is run by tools/r06_natural.sh: r06_clip_*)
  set -- $v
  run $1 $2 qpsk_8_9   --mod-cod QPSK-S_8/9   -m 3.6 -M 3.81 -s 0.1
  run $1 $2 qpsk_3_5   --mod-cod QPSK-S_3/5   -m 1.3 -M 1.51 -s 0.1
  run $1 $2 8psk_3_5   --mod-cod 8PSK-S_3/5   -m 2.7 -M 3.01 -s 0.1
  run $1 $2 8psk_8_9   --mod-cod 8PSK-S_8/9   -m 6.2 -M 6.51 -s 0.1
  run $1 $2 16apsk_8_9 --mod-cod 16APSK-S_8/9 -m 7.1 -M 7.51 -s 0.1 --est-type PERFECT
  echo "== $1"; grep -hE "^ +[0-9]" $OUT/r06_${1}${SFX}_*.txt
done
if [ -z "$NO_PAIRED" ]; then
: > $OUT/r06_paired${SFX}.txt
paired() { modcod=$1; eb=$2; frames=$3; shift 3
  for implem in SPA_EXACT SPA_TANH SPA; do      # (results/r06/r06_paired.txt / r06_paired3.txt were made when the exact rule was still called SPA and the capped one SPA_CLIP)
    l=$(./host/dvbs2_tx_rx_bb --mod-cod $modcod -m $eb -M $(python3 -c "print($eb + 0.01)") -s 0.1 --dec-implem $implem --dec-ite 50 -F 8192 -e 100000000 --max-frames $frames "$@" $EXTRA | grep -E "^ +[0-9]")
    echo "$modcod $eb $implem $l" >> $OUT/r06_paired${SFX}.txt
  done; }
paired QPSK-S_8/9 3.6 49152; paired QPSK-S_8/9 3.7 98304; paired QPSK-S_8/9 3.8 786432
paired QPSK-S_3/5 1.3 49152; paired QPSK-S_3/5 1.4 98304; paired QPSK-S_3/5 1.5 1081344
paired 8PSK-S_3/5 2.7 49152; paired 8PSK-S_3/5 2.8 49152; paired 8PSK-S_3/5 2.9 147456; paired 8PSK-S_3/5 3.0 983040
paired 8PSK-S_8/9 6.2 49152; paired 8PSK-S_8/9 6.3 49152; paired 8PSK-S_8/9 6.4 147456; paired 8PSK-S_8/9 6.5 884736
paired 16APSK-S_8/9 7.1 49152 --est-type PERFECT; paired 16APSK-S_8/9 7.2 49152 --est-type PERFECT; paired 16APSK-S_8/9 7.3 98304 --est-type PERFECT
paired 16APSK-S_8/9 7.4 344064 --est-type PERFECT; paired 16APSK-S_8/9 7.5 2949120 --est-type PERFECT
cat $OUT/r06_paired${SFX}.txt
fi
