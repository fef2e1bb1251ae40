cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
( cd host && make -s ) 2>&1 | tail -2
timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -k "spa_tanh" 2>&1 | tail -5 | tee $OUT/r06_g2_pytest.txt
timeout 300 python tools/noise_moments.py 2>&1 | tail -3 | tee $OUT/r06_noise_moments.txt
: > $OUT/r06_paired3.txt
paired() { modcod=$1; eb=$2; frames=$3; shift 3
  for implem in SPA SPA_TANH SPA_CLIP; do
    l=$(./host/dvbs2_tx_rx_bb --mod-cod $modcod -m $eb -M $(python3 -c "print($eb + 0.01)") -s 0.1 --dec-implem $implem --dec-ite 50 -F 8192 -e 100000000 --max-frames $frames "$@" | grep -E "^ +[0-9]")
    echo "$modcod $eb $implem $l" >> $OUT/r06_paired3.txt
  done; }
paired QPSK-S_3/5 1.4 196608; paired QPSK-S_3/5 1.5 2162688
paired 8PSK-S_3/5 2.9 294912; paired 8PSK-S_3/5 3.0 1966080
paired QPSK-S_8/9 3.8 1572864; paired 16APSK-S_8/9 7.5 2949120 --est-type PERFECT
cat $OUT/r06_paired3.txt
