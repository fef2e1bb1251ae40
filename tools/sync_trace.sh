#!/bin/bash
# GPU box: kernel-trace stats of the synchronizer workload only (tools/pmc_workload.py sync) -> gpurun_out/sy_stats.txt
set -u
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/sy_stats"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/sy_stats" -- python3 "$REPO/tools/pmc_workload.py" sync > "$OUT/sy_stats.log" 2>&1
python3 - "$OUT"/sy_stats/*/*_kernel_trace.csv <<'PY' | tee "$OUT/sy_stats.txt"
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s grid %-10s n %-3d avg %8.1f us  min %8.1f" % (k, g, len(v), sum(v) / len(v), min(v)))
PY
