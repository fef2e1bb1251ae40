#!/bin/bash
# Same-box A/B of prebuilt library variants (tools/build_variant.sh; DVBS2HIP_LIB selects one): runs $AB_CMD (default: the SPA
# timing of tools/bench_spa.py) for every tools/bin/lib_*.so and the in-tree library, $AB_ROUNDS times round-robin.
cd "${GRAFT_REPO_ROOT:-.}"
CMD="${AB_CMD:-python tools/bench_spa.py}"
for i in $(seq 1 ${AB_ROUNDS:-2}); do
  for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_*.so; do
    [ -f "$lib" ] || continue
    echo "== round $i  $lib"
    DVBS2HIP_LIB=$PWD/$lib bash -c "$CMD" 2>&1 | grep -v amdgpu.ids | grep -E "${AB_GREP:-.}"
  done
done
