cd "${GRAFT_REPO_ROOT:-.}"; OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_ldpc_gpu.py -q -k "spa" 2>&1 | tail -8 | tee $OUT/r06_g12_pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/r06_g12_spa_ab.txt
import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
dev = torch.device("cuda", 0)
for modcod, F in (("QPSK-S_8/9", 8192), ("QPSK-S_3/5", 8192), ("32APSK-S_3/4", 8192), ("QPSK-N_8/9", 4096)):
    for rnd in range(2):
        for implem in ("SPA", "SPA_EXACT"):
            rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False, implem=implem)
            g = torch.Generator(device=dev); g.manual_seed(3)
            sg = 0.42
            x = (1.0 + sg * torch.randn((F, rx.N_ldpc), generator=g, device=dev)) * (2.0 / sg ** 2)
            c, b = torch.empty((F,), dtype=torch.int8, device=dev), torch.empty((F, rx.K_ldpc), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for _ in range(2): rx.decode_siho_dev(x.data_ptr(), c.data_ptr(), b.data_ptr(), F)
            rx.synchronize(); rx.timing_enable(True); rx.timing_reset()
            for _ in range(7): rx.decode_siho_dev(x.data_ptr(), c.data_ptr(), b.data_ptr(), F)
            rx.synchronize(); ms, k = rx.timing_get(B.K_LDPC)
            print("%-13s %-9s %s  %.3f ms  %.0f k frames/s  cwd %d" % (modcod, implem, rx.ldpc_kernel_name(), ms / k, F / (ms / k), int(c.sum())), flush=True)
            rx.close()
PY
