#!/bin/bash
# The GPU passes of a round, on the GPU box (gpurun -- 'bash tools/gpu_round.sh STAGE...'); outputs under gpurun_out/<stage>_*.
#   tests     the whole -m gpu suite (builds host/ first)            bench      bench.py line (+ CPU baseline) -> bench.json
#   kernels   tools/bench_kernels.py per-kernel table                rccl       1-rank RCCL logs (torchrun bench.py, C++ simulator)
#   prof      rocprofv3 passes of bench.py (LDPC kernel, tools/summarize_profiles.py afterwards)
#   profk     rocprofv3 passes of tools/pmc_workload.py (every other kernel, tools/summarize_kernels_pmc.py afterwards)
#   refs      smoke() + tools/compare_refs.py + tools/run_ber_sweeps.sh      floor      tools/gpu_floor.sh (tens of millions of frames per point)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/hostmake.log 2>&1
for stage in "$@"; do case $stage in
  tests)   timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests_pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/tests_pytest.log ;;
  bench)   python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json ;;
  kernels) python tools/bench_kernels.py $OUT/kernels.json > $OUT/kernels.log 2>&1; tail -2 $OUT/kernels.log ;;
  rccl)    NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/torchrun_1rank.log 2>&1
           NCCL_DEBUG=INFO DVBS2HIP_FORCE_RCCL=1 ./host/dvbs2_tx_rx_bb --mod-cod QPSK-N_8/9 -m 3.9 -M 4.01 -s 0.1 --dec-implem NMS --dec-ite 10 -F 4096 --world 1 --rank 0 --max-frames 200000 > $OUT/cpp_rccl_1rank.log 2>&1; tail -3 $OUT/cpp_rccl_1rank.log ;;
  prof)    bash tools/profile_gpu.sh > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log ;;
  profk)   bash tools/profile_kernels.sh > $OUT/profile_kernels.log 2>&1; tail -2 $OUT/profile_kernels.log ;;
  refs)    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
           python tools/compare_refs.py $OUT/refs_comparison.md > $OUT/refs_comparison.log 2>&1; tail -2 $OUT/refs_comparison.md
           bash tools/run_ber_sweeps.sh > $OUT/ber_sweeps.log 2>&1; tail -3 $OUT/ber_sweeps.log ;;
  floor)   bash tools/gpu_floor.sh ;;
  *) echo "unknown stage $stage" ;;
esac; done
