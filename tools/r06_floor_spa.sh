#!/bin/bash
# the capped sum-product rule (`--dec-implem SPA`) far down the curves: does the cap leave a floor?  (beside SPA_EXACT on the same seeds) -> gpurun_out/r06_floor_spa.txt
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
run() { echo "== $*"; python -m dvbs2_amd.sim "$@" --clones 1 2>&1 | grep -E "^ +[0-9]"; }
{ for implem in SPA SPA_EXACT; do
  run --mod-cod QPSK-N_8/9 -m 3.80 -M 3.91 -s 0.1 --dec-implem $implem --dec-ite 50 -F 4096 --max-frames 10000000 -e 100000000
  run --mod-cod QPSK-S_8/9 -m 4.10 -M 4.31 -s 0.1 --dec-implem $implem --dec-ite 50 -F 8192 --max-frames 20000000 -e 100000000
  run --mod-cod QPSK-S_3/5 -m 1.70 -M 1.81 -s 0.1 --dec-implem $implem --dec-ite 50 -F 8192 --max-frames 20000000 -e 100000000
done; } 2>&1 | tee $OUT/r06_floor_spa.txt
