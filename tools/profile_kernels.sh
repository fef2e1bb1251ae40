#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes over tools/pmc_workload.py (the non-LDPC
# kernels of the path).  tools/summarize_kernels_pmc.py condenses the outputs into profiles/<tag>_kernels_pmc.md.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="$REPO/tools/pmc_workload.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pk_stats" -- python3 $W > "$OUT/pk_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pk_fetch" -- python3 $W > "$OUT/pk_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pk_write" -- python3 $W > "$OUT/pk_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pk_sq1" -- python3 $W > "$OUT/pk_sq1.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pk_sq2" -- python3 $W > "$OUT/pk_sq2.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d "$OUT/pk_sq3" -- python3 $W > "$OUT/pk_sq3.log" 2>&1
tail -2 "$OUT/pk_stats.log"; ls "$OUT"/pk_*/*/ | head
