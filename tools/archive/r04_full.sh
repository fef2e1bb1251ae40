#!/bin/bash
# Round 4 full pass on the GPU box: the whole -m gpu suite, the bench line, the LDPC profile passes, the issue probes
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/hostmake.log 2>&1
tools/bin/probe_issue > $OUT/r04_probe_issue.txt 2>&1; tools/bin/probe_issue2 > $OUT/r04_probe_issue2.txt 2>&1; tools/bin/probe_addr > $OUT/r04_probe_addr.txt 2>&1
timeout 3000 python -m pytest tests -m gpu -q -x > $OUT/r04_tests.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/r04_tests.log
python bench.py --steps 20 --warmup 3 > $OUT/r04_bench.json 2> $OUT/r04_bench.err; tail -c 600 $OUT/r04_bench.json; tail -3 $OUT/r04_bench.err
bash tools/profile_gpu.sh > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
