#!/bin/bash
# round 5: the reference's own configuration (QPSK-S 8/9, SPA 50 ite, early stop) at 3.8 dB, -F 8192: kernel-trace stats of the whole TX -> AWGN -> RX loop
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"; export TMPDIR=/tmp
( cd host && make -s ) 2>&1 | tail -2
F=${1:-8192}
./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.8 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F $F 2>&1 | grep -E "^ +[0-9]"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_refcfg" -- ./host/dvbs2_tx_rx_bb --mod-cod QPSK-S_8/9 -m 3.8 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F $F > "$OUT/prof_refcfg.log" 2>&1
f=$(ls -t $OUT/prof_refcfg/*/*_kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-70s calls %5s avg %10.1f us  total %9.2f ms %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
t=$(ls -t $OUT/prof_refcfg/*/*_kernel_trace.csv | head -1); python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("span %.2f ms, kernels busy %.2f ms (%.1f %%), %d launches" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len(rows)))
l = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "ldpc" in r["Kernel_Name"] and "enc" not in r["Kernel_Name"].lower()]
print("ldpc decode launches (us):", " ".join("%.0f" % x for x in l))
PY
grep -E "^ +[0-9]" "$OUT/prof_refcfg.log"
