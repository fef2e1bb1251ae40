#!/bin/bash
# Same-box A/B of kernel sources (fresh gpurun boxes differ by 2-3 % in absolute time, more than most single changes are
# worth): the working tree vs older versions of some csrc files placed in tools/ab_old/ (e.g. `git show <rev>:dvbs2_amd/csrc/
# k_ldpc_wg8.hip > tools/ab_old/k_ldpc_wg8.hip`; the directory is git-ignored).  GPU box only: builds both libraries there and
# alternates them three times.  AB_MODE = bench (default: bench.py, ms per step) | sim (early-stop simulator, Mb/s) | cmd
# (runs $AB_CMD, which prints its own line).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
cp dvbs2_amd/lib/libdvbs2hip.so /tmp/lib_new.so
mkdir -p /tmp/ab_new
for f in tools/ab_old/*; do b=$(basename $f); cp dvbs2_amd/csrc/$b /tmp/ab_new/$b; cp $f dvbs2_amd/csrc/$b; done
python -c "from dvbs2_amd import build; build.build_lib(force=True)" > /dev/null 2>&1
cp dvbs2_amd/lib/libdvbs2hip.so /tmp/lib_old.so
for f in /tmp/ab_new/*; do cp $f dvbs2_amd/csrc/$(basename $f); done
for i in 1 2 3; do for v in old new; do
  cp /tmp/lib_$v.so dvbs2_amd/lib/libdvbs2hip.so
  case "${AB_MODE:-bench}" in
    sim) python -m dvbs2_amd.sim --mod-cod QPSK-N_8/9 -m 3.9 -M 4.11 -s 0.1 --dec-implem NMS --dec-ite 10 -F 2048 --max-frames 400000 2>/dev/null | grep -E "^ +[0-9]" | awk -v v=$v '{printf "%s Eb/N0 %s  FE %s  %s Mb/s\n", v, $3, $9, $15}' ;;
    cmd) echo -n "$v "; bash -c "$AB_CMD" 2>/dev/null | tail -1 ;;
    *)   python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), d['ber']['BE'])" ;;
  esac
done; done
cp /tmp/lib_new.so dvbs2_amd/lib/libdvbs2hip.so
