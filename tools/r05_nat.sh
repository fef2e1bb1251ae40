#!/bin/bash
# round 5: natural row order: the side-by-side kernel's workgroup shapes (checks side by side x waves per workgroup = frames per image row) by batch size
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$REPO"
for parts in ${PARTS:-44 88 82}; do echo "parts $parts: "; DVBS2HIP_NAT_PARTS=$parts timeout 600 python tools/bench_natural.py ${@:-4096} 2>&1 | grep natural | sed 's/ldpc_nat_kernel.*batch size//'; done
