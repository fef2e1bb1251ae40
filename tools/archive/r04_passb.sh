#!/bin/bash
# Round 4 pass B on the GPU box: the bench line with the stamped traffic, 1-rank RCCL logs, the whole -m gpu suite; fabric traffic of mode 6 for the record
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT/final
bash tools/gpu_final_pass.sh B
REPO=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  DVBS2HIP_LDPC_FAST_MODE=cu1 rocprofv3 --pmc $c --output-format csv -d "$REPO/$OUT/cu1_$c" -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --self-check-steps 0 > "$REPO/$OUT/cu1_$c.log" 2>&1
done
cd $REPO; python - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/cu1_%s/*/*counter_collection.csv" % c):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "ldpc" in r["Kernel_Name"] and r["Counter_Name"] == c]
        if v: print("mode 6", c, "KiB per launch:", sum(v) / len(v), "kernel", [r["Kernel_Name"] for r in csv.DictReader(open(f)) if "ldpc" in r["Kernel_Name"]][0][:60])
PY
