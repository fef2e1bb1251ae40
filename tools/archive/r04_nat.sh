#!/bin/bash
# natural row order: parity of every form, then the BASELINE batch with a check's edges over 8 lanes (8) against eight consecutive checks side by side (88)
cd "${GRAFT_REPO_ROOT:-.}"
timeout 1200 python -m pytest tests/test_ldpc_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "natural or golden" 2>&1 | tail -1
for i in 1 2; do for parts in 81 88 44; do DVBS2HIP_NAT_PARTS=$parts timeout 300 python tools/bench_natural.py 2>&1 | grep natural | sed "s/^/parts $parts: /"; done; done
