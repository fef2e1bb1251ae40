#!/bin/bash
# CPU container: the timing-only ablation builds of the min-sum layer (-DW8_ABL=..., wrong results by construction) for tools/run_ablations.sh:
#   1 = pass 2's global stores dropped, 2 = pass 1a's global loads replaced by a register move, 4 = pass 1b without the min / sign tracking,
#   8 = pass 2 without the compare and the selects; 3 = no global slot traffic, 12 = both arithmetic cuts, 15 = everything above (the dependent chain alone)
set -e
cd "$(dirname "$0")/.."
rm -f tools/bin/lib_abl*.so tools/bin/k_ldpc_wg8_abl*.o
# (one after the other: seven hipcc runs of this file at once lose some of them in this 8-CPU container)
for a in 1 2 3 4 8 12 15; do bash tools/build_variant.sh abl$a k_ldpc_wg8 -DW8_ABL=$a > /dev/null 2>&1; test -f tools/bin/lib_abl$a.so || { echo "lib_abl$a.so did not build"; exit 1; }; done
ls tools/bin/lib_abl*.so
