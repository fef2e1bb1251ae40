cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
DVBS2HIP_LIB=$PWD/tools/bin/lib_prof.so python - <<'PY' 2>&1 | grep -v amdgpu | tail -30 | tee gpurun_out/r06_g10_phase.txt
import os, sys, torch
sys.path.insert(0, os.getcwd())
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
for modcod in ("32APSK-S_3/4", "QPSK-S_8/9"):
    rx = Dvbs2Hip(modcod, max_frames=1, n_ite=10, alpha=1.0, early_stop=False)
    x = 8.0 * (1.0 + 0.3 * torch.randn((1, rx.N_ldpc), device=dev))
    c, b = torch.empty((1,), dtype=torch.int8, device=dev), torch.empty((1, rx.K_ldpc), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    print("==", modcod, rx.ldpc_kernel_name(), flush=True)
    for _ in range(2):
        rx.decode_siho_dev(x.data_ptr(), c.data_ptr(), b.data_ptr(), 1); rx.synchronize()
    rx.close()
PY
