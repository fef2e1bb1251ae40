#!/bin/bash
# tools/sync_in_loop.py on the reference's other MODCODs at one point of each one's waterfall, with the baseband loop at the same point beside it -> gpurun_out/sync_in_loop_<modcod>.*
cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; mc=$2; e=$3; shift 3
  python -m dvbs2_amd.sim --mod-cod $mc -m $e -M $(python -c "print($e + 0.01)") --dec-implem SPA --dec-ite 50 -F 2048 -e 600 --max-frames 400000 --clones 2 "$@" --json $OUT/sync_in_loop_${name}_bb.json 2>&1 | grep -E "^ +[0-9]"
  est=""; [ "$1" = "--est-type" ] && est="--est-perfect"
  timeout 600 python tools/sync_in_loop.py --mod-cod $mc --ebn0 $e --fe 300 --max-frames 80000 --freq 1e-4 --skip 64 $est --json $OUT/sync_in_loop_$name.json 2>&1 | grep -v amdgpu; }
run 8psk_3_5 8PSK-S_3/5 2.9
run 8psk_8_9 8PSK-S_8/9 6.4
run 16apsk_8_9 16APSK-S_8/9 7.4 --est-type PERFECT
run qpsk_3_5 QPSK-S_3/5 1.5
