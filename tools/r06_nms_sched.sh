#!/bin/bash
# NMS, 10 iterations (the BASELINE decoder, the reference README's `--dec-implem NMS --dec-ite 10`): QC layers against the reference's natural row order on the SAME frames,
# and QC with 11 / 12 iterations -> gpurun_out/r06_nms_sched.txt
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
: > $OUT/r06_nms_sched.txt
paired() { modcod=$1; eb=$2; frames=$3; F=$4
  for v in "QC 10" "NATURAL 10" "QC 11" "QC 12"; do set -- $v
    l=$(timeout 600 ./host/dvbs2_tx_rx_bb --mod-cod $modcod -m $eb -M $(python3 -c "print($eb + 0.01)") -s 0.1 --dec-implem NMS --dec-sched $1 --dec-ite $2 -F $F --clones 2 -e 100000000 --max-frames $frames | grep -E "^ +[0-9]")
    echo "$modcod $eb NMS $1 ite $2 $l" >> $OUT/r06_nms_sched.txt
  done; }
paired QPSK-S_8/9 4.0 262144 16384; paired QPSK-S_8/9 4.3 1048576 16384
paired QPSK-N_8/9 3.9 131072 8192; paired QPSK-N_8/9 4.0 524288 8192
cat $OUT/r06_nms_sched.txt
