#!/bin/bash
# round-2 GPU pass F: smoke(), the refs comparison table and the BER/FER sweeps of every MODCOD with the round-2 build
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
python tools/compare_refs.py $OUT/f_refs_comparison.md > $OUT/f_refs_comparison.log 2>&1; tail -2 $OUT/f_refs_comparison.md
bash tools/run_ber_sweeps.sh > $OUT/f_ber_sweeps.log 2>&1; tail -3 $OUT/f_ber_sweeps.log
