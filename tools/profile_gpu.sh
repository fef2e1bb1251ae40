#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of
# `bench.py` (same command the bench line comes from).  Outputs under gpurun_out/prof_*.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --self-check-steps 0 --self-check-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 $ARGS > "$OUT/prof_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -- python3 $ARGS > "$OUT/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -- python3 $ARGS > "$OUT/prof_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/prof_sq1" -- python3 $ARGS > "$OUT/prof_sq1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/prof_sq2" -- python3 $ARGS > "$OUT/prof_sq2.log" 2>&1
find "$OUT" -name "*.csv" | head -40
tail -2 "$OUT/prof_stats.log"
# calibration of FETCH_SIZE / WRITE_SIZE on a known dword-per-lane copy (1.06 GB read + 1.06 GB written per launch)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_cal_fetch" -- python3 $REPO/tools/calibrate_fetch.py > "$OUT/prof_cal_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_cal_write" -- python3 $REPO/tools/calibrate_fetch.py > "$OUT/prof_cal_write.log" 2>&1
