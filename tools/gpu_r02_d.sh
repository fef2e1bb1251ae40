#!/bin/bash
# round-2 GPU pass D: GPU test-suite, bench line (bounded roofline from the committed PMC file), per-kernel table, 1-rank RCCL logs.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/d_hostmake.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $OUT/d_pytest.log 2>&1; echo "pytest rc $?"; tail -4 $OUT/d_pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/d_bench.json 2> $OUT/d_bench.err; tail -c 300 $OUT/d_bench.json
python tools/bench_kernels.py $OUT/d_kernels.json > $OUT/d_kernels.log 2>&1; tail -2 $OUT/d_kernels.log
NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/d_torchrun_1rank.log 2>&1
NCCL_DEBUG=INFO DVBS2HIP_FORCE_RCCL=1 ./host/dvbs2_tx_rx_bb --mod-cod QPSK-N_8/9 -m 3.9 -M 4.01 -s 0.1 --dec-implem NMS --dec-ite 10 -F 4096 --world 1 --rank 0 --max-frames 200000 > $OUT/d_cpp_rccl_1rank.log 2>&1; tail -3 $OUT/d_cpp_rccl_1rank.log
bash tools/profile_kernels.sh > $OUT/d_profile_kernels.log 2>&1
