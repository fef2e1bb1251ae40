#!/bin/bash
# Same-box A/B of the LDPC kernel (fresh gpurun boxes differ by 2-3 % in absolute time, more than most single changes are
# worth): the working tree vs older sources placed in tools/ab_old/ (e.g. `git show <rev>:dvbs2_amd/csrc/k_ldpc_wg8.hip >
# tools/ab_old/k_ldpc_wg8.hip`, same for k_ldpc.hip; the directory is git-ignored).  GPU box only: builds both libraries there
# and alternates them, 3 x 20 launches each.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
cp dvbs2_amd/lib/libdvbs2hip.so /tmp/lib_new.so
cp dvbs2_amd/csrc/k_ldpc_wg8.hip /tmp/new_wg8.hip; cp dvbs2_amd/csrc/k_ldpc.hip /tmp/new_ldpc.hip
cp tools/ab_old/k_ldpc_wg8.hip dvbs2_amd/csrc/k_ldpc_wg8.hip; cp tools/ab_old/k_ldpc.hip dvbs2_amd/csrc/k_ldpc.hip
python -c "from dvbs2_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
cp dvbs2_amd/lib/libdvbs2hip.so /tmp/lib_old.so
cp /tmp/new_wg8.hip dvbs2_amd/csrc/k_ldpc_wg8.hip; cp /tmp/new_ldpc.hip dvbs2_amd/csrc/k_ldpc.hip
for i in 1 2 3; do for v in old new; do
  cp /tmp/lib_$v.so dvbs2_amd/lib/libdvbs2hip.so
  if [ "${AB_MODE:-bench}" = sim ]; then     # early-stop workload: the Monte-Carlo simulator (SIM_THR column, Mb/s)
    python -m dvbs2_amd.sim --mod-cod QPSK-N_8/9 -m 3.9 -M 4.11 -s 0.1 --dec-ite 10 -F 2048 --max-frames 400000 2>/dev/null | grep -E "^ +[0-9]" | awk -v v=$v '{printf "%s Eb/N0 %s  FE %s  %s Mb/s\n", v, $3, $9, $15}'
  else
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d['ber']['BE'])"
  fi
done; done
