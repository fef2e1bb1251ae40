cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r3h
export DVBS2HIP_SPA_MPITCH=1440
for i in 1 2 3 4 5 6; do python tools/bench_spa.py 2>&1 | grep "N_8/9"; done > gpurun_out/r3h/var.txt 2>&1; cat gpurun_out/r3h/var.txt
rocm-smi --showmeminfo vram 2>&1 | tail -4
