#!/bin/bash
# rocprofv3 passes over tools/fir_only.py (kernel stats, then PMC groups in separate runs). Outputs: gpurun_out/prof_fir_*
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_fir_stats" -- python3 $REPO/tools/fir_only.py > "$OUT/prof_fir_stats.log" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/prof_fir_pmc$i" -- python3 $REPO/tools/fir_only.py > "$OUT/prof_fir_pmc$i.log" 2>&1
done
python3 - <<'P'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out"
for f in sorted(glob.glob(out + "/prof_fir_stats/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fir" in r["Name"]: print("stats", r["Name"][:60], r["Calls"], r["AverageNs"])
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/prof_fir_pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fir_mfma" in r["Kernel_Name"] or "fir_ccr" in r["Kernel_Name"]:
            a = acc[(r["Kernel_Name"][:40], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
# one row per dispatch and counter (already summed over dimensions?) -- print per-dispatch averages (5 dispatches per run)
for (k, c), (v, n) in sorted(acc.items()): print("pmc", k, c, "%.4g total over run, %.4g per launch" % (v, v / 5))
P
