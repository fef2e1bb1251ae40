#!/bin/bash
# kernel-trace durations of the frame synchronizer's kernels for the in-tree library and every variant under tools/bin (32APSK-S 4096 frames, QPSK-N 1024 frames)
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in $REPO/dvbs2_amd/lib/libdvbs2hip.so $(ls $REPO/tools/bin/lib_*.so 2>/dev/null | sort); do
  for c in "32APSK-S_3/4 4096" "QPSK-N_8/9 1024"; do
    rm -rf "$OUT/sy_abl"
    DVBS2HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d "$OUT/sy_abl" -- python3 "$REPO/tools/sync_time.py" $c > "$OUT/sy_abl.log" 2>&1
    python3 - "$(basename $lib)" "$c" "$OUT"/sy_abl/*/*_kernel_trace.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[3])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dvbs2::", "")
    if k.startswith("sync_"): acc[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[1], sys.argv[2], " ".join("%s %.1f" % (k[5:25], sorted(v)[len(v) // 2]) for k, v in sorted(acc.items())))
PY
  done
done
