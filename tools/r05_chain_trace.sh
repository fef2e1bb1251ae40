#!/bin/bash
# round 5: timeline of one fused-chain call (dvbs2hip_rx_bb_dev, BASELINE configs[2] / [3]): every kernel of the call, its duration and the gap in front of it
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mc in ${1:-QPSK-N_8/9} ${2:-16APSK-N_8/9}; do
  n=10; [ "$mc" = "16APSK-N_8/9" ] && n=20
  d="$OUT/chain_trace_$(echo $mc | tr '/' '_')"; rm -rf "$d"
  rocprofv3 --kernel-trace --output-format csv -d "$d" -- python3 "$REPO/tools/chain_time.py" $mc 4096 $n > "$d.log" 2>&1
  tail -1 "$d.log"
  python3 - "$d"/*/*_kernel_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "ldpc_" in r["Kernel_Name"] and "enc" not in r["Kernel_Name"]]
last = idx[-1]
# one whole call = from behind the previous LDPC launch to this one's followers
lo = idx[-2] + 1
while lo < last and not ("front" in rows[lo]["Kernel_Name"] or "sync" in rows[lo]["Kernel_Name"]): lo += 1
prev = None; t0 = int(rows[lo]["Start_Timestamp"])
for i in range(lo, min(len(rows), last + 6)):
    r = rows[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  %-60s grid %-9s start +%9.1f us  dur %8.1f us  gap %6.1f us" % (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"], (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    prev = e
PY
done
