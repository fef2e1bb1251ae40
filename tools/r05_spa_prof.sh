#!/bin/bash
# round 5: the sum-product mode-6 kernel: per-role phase profile (tools/bin/lib_spaprof.so) and PMC passes (fabric bytes, VALU instructions, busy cycles)
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$REPO"
DVBS2HIP_LIB=$REPO/tools/bin/lib_spaprof.so DVBS2HIP_LDPC_FAST_MODE=cu1 timeout 600 python tools/bench_spa.py 4096 0 1 2>&1 | grep -v amdgpu.ids | tail -40
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  d=$REPO/gpurun_out/r05_spa_pmc_$(echo $c | cut -d' ' -f1)
  DVBS2HIP_LDPC_FAST_MODE=cu1 rocprofv3 --pmc $c --output-format csv -d $d -- python3 $REPO/tools/bench_spa.py 4096 0 2 > $d.log 2>&1
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ldpc" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "avg per launch %.6g over %d launches" % (sum(v) / len(v), len(v)))
PY
done
