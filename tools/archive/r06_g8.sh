cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_sync_gpu.py tests/test_rx_lite_gpu.py -q 2>&1 | tail -8 | tee $OUT/r06_g8_pytest.txt
for s in 1 8; do echo "DVBS2HIP_SYNC_SEGMENTS=$s"; DVBS2HIP_SYNC_SEGMENTS=$s python tools/sync_located_time.py 32APSK-S_3/4 4096 50 2>&1 | grep -v amdgpu; DVBS2HIP_SYNC_SEGMENTS=$s python tools/sync_located_time.py QPSK-S_8/9 4096 50 2>&1 | grep -v amdgpu;  DVBS2HIP_SYNC_SEGMENTS=$s python tools/sync_located_time.py 16APSK-S_8/9 8192 50 2>&1 | grep -v amdgpu; done 2>&1 | tee $OUT/r06_g8_sync.txt
