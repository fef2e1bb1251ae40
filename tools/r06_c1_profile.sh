#!/bin/bash
# Where one clone's batch goes (QPSK-S 8/9, 3.8 dB, SPA 50 iterations with the stopping rule, -F 8192): the kernels of a fixed number of batches -> gpurun_out/r06_c1_stats.txt
cd "${GRAFT_REPO_ROOT:-.}"; OUT=$PWD/gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
export TMPDIR=/tmp
ARGS="--mod-cod QPSK-S_8/9 -m 3.8 -M 3.81 -s 0.1 --dec-implem SPA --dec-ite 50 -F 8192 --clones 1 -e 100000000 --max-frames 819200"
./host/dvbs2_tx_rx_bb $ARGS | grep -E "^ +[0-9]"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c1prof -o c1 -- ./host/dvbs2_tx_rx_bb $ARGS > $OUT/r06_c1_run.txt 2>&1
f=$(find $OUT/c1prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $OUT/r06_c1_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per 100 batches of 8192 frames: %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%6.2f %%  %8.3f ms/batch  calls %5s  avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6 / 100, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
rm -rf $OUT/c1prof
