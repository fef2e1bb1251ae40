#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the LDPC kernel's instantiations that the
# reference's own configuration uses (SPA, N = 16200) and the short-frame NMS kernels -- VERDICT r2 item 1.  tools/summarize_ldpc_variants.py
# condenses gpurun_out/lv_* into profiles/<tag>_ldpc_variants.md.
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="$REPO/tools/pmc_ldpc_workload.py"
i=0
while read -r modcod implem frames; do
  [ -z "$modcod" ] && continue
  i=$((i+1)); tag="lv_${i}"
  echo "$modcod $implem $frames" > "$OUT/$tag.cfg"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${tag}_stats" -- python3 $W $modcod $implem $frames > "$OUT/${tag}_stats.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${tag}_fetch" -- python3 $W $modcod $implem $frames > "$OUT/${tag}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${tag}_write" -- python3 $W $modcod $implem $frames > "$OUT/${tag}_write.log" 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/${tag}_sq1" -- python3 $W $modcod $implem $frames > "$OUT/${tag}_sq1.log" 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/${tag}_sq2" -- python3 $W $modcod $implem $frames > "$OUT/${tag}_sq2.log" 2>&1
  tail -1 "$OUT/${tag}_stats.log"
done <<CFG
QPSK-N_8/9 NMS 4096
QPSK-N_8/9 SPA 4096
QPSK-S_8/9 NMS 8192
QPSK-S_8/9 SPA 8192
QPSK-S_8/9 SPA_TANH 8192
QPSK-S_3/5 NMS 8192
QPSK-S_3/5 SPA 8192
32APSK-S_3/4 NMS 8192
CFG
