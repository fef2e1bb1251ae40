cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
for rnd in 1 2; do for lib in dvbs2_amd/lib/libdvbs2hip.so tools/bin/lib_bs2.so tools/bin/lib_bs1.so; do
 echo "$(basename $lib): $(DVBS2HIP_LIB=$PWD/$lib python tools/bench_spa.py 8192 8192 3 2>&1 | grep 'QPSK-S_8/9 8192 SPA' | sed 's/.*SPA //')"
done; done 2>&1 | tee $OUT/r06_g14_bs.txt
