cd "${GRAFT_REPO_ROOT:-.}"
for lib in tools/bin/lib_d07.so tools/bin/lib_d17.so; do echo "== $lib"; DVBS2HIP_LIB=$PWD/$lib python tools/spa_dbg.py 2>&1 | grep -v amdgpu | cut -c1-60; done
