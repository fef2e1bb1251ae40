cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
python -m dvbs2_amd.sim --mod-cod QPSK-N_8/9 -m 3.9 -M 4.31 -s 0.1 --dec-implem NMS --dec-ite 10 -F 4096 --max-frames 1500000 2>/dev/null | grep -E "^ +[0-9]|Es/N0" | tee gpurun_out/ab/sim_qpsk_n.txt
python -m dvbs2_amd.sim --mod-cod QPSK-S_8/9 -m 4.0 -M 4.41 -s 0.2 --dec-implem NMS --dec-ite 10 -F 8192 --max-frames 4000000 2>/dev/null | grep -E "^ +[0-9]" | tee gpurun_out/ab/sim_qpsk_s.txt
python -m dvbs2_amd.sim --mod-cod 16APSK-N_8/9 -m 7.8 -M 8.01 -s 0.2 --dec-implem NMS --dec-ite 20 -F 4096 --max-frames 800000 2>/dev/null | grep -E "^ +[0-9]" | tee gpurun_out/ab/sim_16apsk_n.txt
