#!/bin/bash
# The filtered loop of the reference's dvbs2_tx_rx with genie timing (TX mirror -> shaping filter -> AWGN at the sample rate -> matched filter -> extraction -> RX chain),
# beside the baseband loop at the same Eb/N0: with unit-energy RRC taps on both sides the two are the same channel, so their FER must agree; the reference's traces for
# this loop (refs/TX_RX/*.txt) run its sample-serial synchronizers instead of a genie and lie above both.  -> gpurun_out/r06_filtered_*.txt
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
for mode in filtered bb; do
  flag=""; [ $mode = filtered ] && flag="--filtered"
  python -m dvbs2_amd.sim --mod-cod QPSK-S_8/9 -m 3.4 -M 3.91 -s 0.1 --dec-implem SPA --dec-ite 50 -F 4096 -e 1000 --max-frames 4000000 --clones 3 $flag --json $OUT/r06_filtered_$mode.json 2>&1 | grep -v amdgpu > $OUT/r06_filtered_$mode.txt
  echo "== $mode"; grep -E "^ +[0-9]" $OUT/r06_filtered_$mode.txt
done
