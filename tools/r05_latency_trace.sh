#!/bin/bash
# round 5: BASELINE configs[4] at F = 1: the kernels of one call sequence (matched filter -> extraction -> fused chain), their durations and the gaps between them
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
python3 "$REPO/tools/latency_f1.py" ${1:-1} 40
cd /tmp && export TMPDIR=/tmp
d="$OUT/lat_trace"; rm -rf "$d"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$d" -- python3 "$REPO/tools/latency_f1.py" ${1:-1} 10 > "$d.log" 2>&1
tail -1 "$d.log"
python3 - "$d"/*/*_kernel_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ldpc_" in r["Kernel_Name"] and "tx_" not in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1]
# the last whole sequence: from the first kernel behind the previous LDPC launch's followers
seq = rows[a:b + 4]
t0 = None; prev = None
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print("  %-62s grid %-8s start +%8.1f us  dur %7.1f us  gap %6.1f us" % (r["Kernel_Name"].split("(")[0][-62:], r["Grid_Size_X"], (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
