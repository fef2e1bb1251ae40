#!/bin/bash
# kernel-trace stats of the whole Monte-Carlo loop (TX mirror + RX chain + monitor)
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd "$REPO" && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_sim" -- python3 -m dvbs2_amd.sim --mod-cod ${1:-QPSK-N_8/9} -m 4.0 -M 4.05 --dec-implem NMS --dec-ite 10 -F 4096 --max-frames 40000 > "$OUT/prof_sim.log" 2>&1
f=$(ls -t $OUT/prof_sim/*/*_kernel_stats.csv | head -1); cut -c1-90 $f | head -12; python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-60s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -3 "$OUT/prof_sim.log"
