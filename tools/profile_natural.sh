#!/bin/bash
# GPU box: kernel-trace stats and fabric-byte counters of the natural-row-order decoder at the BASELINE batch (tools/bench_natural_scan.py QPSK-N_8/9 4096) -> gpurun_out/nat_*
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W="$REPO/tools/bench_natural_scan.py QPSK-N_8/9 4096"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/nat_stats" -- python3 $W > "$OUT/nat_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/nat_fetch" -- python3 $W > "$OUT/nat_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/nat_write" -- python3 $W > "$OUT/nat_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/nat_sq" -- python3 $W > "$OUT/nat_sq.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/nat_summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/nat_stats/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-70:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("natural row order, QPSK-N_8/9, 4096 frames, 10 iterations fixed (tools/profile_natural.sh): kernel-trace durations")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "nat" in k: print("  %-70s n %-2d avg %9.1f us  min %9.1f" % (k, len(v), sum(v) / len(v), min(v)))
pm = {}
for d in ("nat_fetch", "nat_write", "nat_sq"):
    for f in glob.glob(out + "/" + d + "/*/*_counter_collection.csv"):
        a = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ldpc_nat" in r["Kernel_Name"]: a[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in a.items(): pm[k] = sum(v) / len(v)
print("counters per launch of the decoder kernel:", {k: "%.4g" % v for k, v in sorted(pm.items())})
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    fb = 2 * pm["FETCH_SIZE"] * 1024 + pm["WRITE_SIZE"] * 1024
    print("fabric bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB counters): %.2f GB; the sweep's own bytes: 4096 frames x 10 iterations x 7200 checks x (27 + 27 + 6) x 4 B = %.2f GB" % (fb / 1e9, 4096 * 10 * 7200 * 60 * 4 / 1e9))
PY
