#!/bin/bash
# round-2 GPU pass A: same-box A/B of the LDPC kernel change, the whole GPU test-suite, the bench line, a 1-rank torchrun log
# (RCCL init + all-reduce), then the rocprofv3 passes of bench.py.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) > $OUT/a_hostmake.log 2>&1
bash tools/ab_kernel.sh > $OUT/a_ab.log 2>&1
tail -6 $OUT/a_ab.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/a_pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/a_pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/a_bench.json 2> $OUT/a_bench.err; tail -c 1500 $OUT/a_bench.json
NCCL_DEBUG=INFO python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/a_torchrun_1rank.log 2>&1
grep -c -i "nccl\|rccl" $OUT/a_torchrun_1rank.log
bash tools/profile_gpu.sh > $OUT/a_profile.log 2>&1
tail -3 $OUT/a_profile.log
