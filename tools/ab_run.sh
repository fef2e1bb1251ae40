cd "${GRAFT_REPO_ROOT:-.}"
python tools/spa_check.py 2>&1 | grep -v amdgpu | head -12 | cut -c1-130
