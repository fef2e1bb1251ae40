cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "reproducible" 2>&1 | tail -3
