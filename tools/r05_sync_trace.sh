#!/bin/bash
# round 5: timeline of the located synchronizer call (kernel trace): kernels, durations and the gaps between them
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/syl_trace"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/syl_trace" -- python3 "$REPO/tools/sync_located_time.py" ${1:-32APSK-S_3/4} ${2:-4096} 6 > "$OUT/syl_trace.log" 2>&1
tail -2 "$OUT/syl_trace.log"
python3 - "$OUT"/syl_trace/*/*_kernel_trace.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the located calls: the 6 timed ones + 3 warm-ups come right after the copy form's; print the last two whole calls before the first front kernel
idx = [i for i, r in enumerate(rows) if "sync_locate_kernel" in r["Kernel_Name"]]
sel = idx[4:7]
start = None
for i in range(sel[0] + 1, sel[-1] + 3):
    r = rows[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if start is None: start = s; prev = s
    print("%-58s grid %-9s wg %-5s start +%8.1f us  dur %7.1f us  gap before %6.1f us" % (r["Kernel_Name"].split("(")[0][-58:], r["Grid_Size_X"], r["Workgroup_Size_X"], (s - start) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
PY
