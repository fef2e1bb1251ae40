#!/bin/bash
# round 5: natural row order at the BASELINE batch, the shapes 8 x 4 (rows of 32 frames, 128 workgroups), 4 x 2 (rows of 32, 128 workgroups) and 8 x 2 (rows of 16, 256 workgroups):
# vector-memory path counters per launch of the decoder kernel (one rocprofv3 --pmc pass per group; no trace domains with --pmc).  NOT the TA_* counters: a pass with
# TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum aborted inside rocprofv3 and hung until the box's limit (25 GPU-minutes)
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="$REPO/tools/bench_natural_scan.py QPSK-N_8/9 4096"
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for parts in 88 44 82; do
    d="$OUT/natpmc_${parts}_$i"; rm -rf "$d"
    DVBS2HIP_NAT_PARTS=$parts timeout 120 rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 $W > "$d.log" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/natpmc_*_*")):
    if d.endswith(".log"): continue
    parts = d.split("natpmc_")[1].split("_")[0]
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ldpc_nat_ck" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in a.items(): res[k][parts] = sum(v) / len(v)
print("natural row order, QPSK-N_8/9, 4096 frames, 10 iterations: counters per launch of ldpc_nat_ck_kernel (tools/r05_nat_pmc.sh)")
print("%-36s %14s %14s %14s" % ("counter", "8 x 4 (88)", "4 x 2 (44)", "8 x 2 (82)"))
for k in sorted(res):
    print("%-36s %14.5g %14.5g %14.5g" % (k, res[k].get("88", float("nan")), res[k].get("44", float("nan")), res[k].get("82", float("nan"))))
PY
for parts in 88 44 82; do DVBS2HIP_NAT_PARTS=$parts python3 $W; done
