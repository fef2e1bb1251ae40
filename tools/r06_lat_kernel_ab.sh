cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -q --maxfail=25 -x 2>&1 | tail -25 > $OUT/r06_lat_pytest.txt; tail -25 $OUT/r06_lat_pytest.txt
for v in 1 0; do echo "DVBS2HIP_LDPC_LAT=$v"; DVBS2HIP_LDPC_LAT=$v python tools/latency_one_frame.py 32APSK-S_3/4 2>&1 | grep -v amdgpu | head -4; DVBS2HIP_LDPC_LAT=$v python tools/latency_one_frame.py QPSK-S_8/9 2>&1 | grep -v amdgpu | head -4; done 2>&1 | tee $OUT/r06_lat_ab.txt
