cd "${GRAFT_REPO_ROOT:-.}"
for it in 2 3 10; do echo "== n_ite $it sigma 0.50"; DET_ITE=$it python tools/det_check.py 4096 0.50 QPSK-S_8/9 QPSK-N_8/9 QPSK-S_3/5 32APSK-S_3/4 2>&1 | grep -v amdgpu | cut -c1-200; done
