#!/bin/bash
# Round 4, item 1(a): per-phase cycle profile (-DLDPC_PHASE_PROF variant, tools/build_variant.sh phase k_ldpc_wg8 -DLDPC_PHASE_PROF) + a PMC pass
# with the wait / level counters of the production kernel.  GPU box only; outputs under gpurun_out/r04_phase_*.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd "$REPO"
for lib in tools/bin/lib_phase*.so; do
  tag=$(basename $lib .so)
  DVBS2HIP_LIB=$REPO/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --self-check-steps 0 > $OUT/r04_${tag}_normal.json 2> $OUT/r04_${tag}_normal.txt
  DVBS2HIP_LIB=$REPO/$lib SCAN_SIZES="16384" python tools/scan_batch.py QPSK-S_8/9 NMS 1 > $OUT/r04_${tag}_short.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --self-check-steps 0"
rocprofv3 -L > "$OUT/r04_counters_avail.txt" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/r04_pmc_a" -- python3 $ARGS > "$OUT/r04_pmc_a.log" 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/r04_pmc_b" -- python3 $ARGS > "$OUT/r04_pmc_b.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_WAVE32_LDS --output-format csv -d "$OUT/r04_pmc_c" -- python3 $ARGS > "$OUT/r04_pmc_c.log" 2>&1
find "$OUT" -name "*counter_collection.csv" | head
tail -30 $OUT/r04_lib_phase_normal.txt
