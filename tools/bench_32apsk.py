import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0); F = 16384
torch.manual_seed(1)
for implem in ("NMS", "SPA"):
    rx = Dvbs2Hip("32APSK-S_3/4", max_frames=F, n_ite=10, alpha=1.0, early_stop=False, implem=implem)
    N, K = rx.N_ldpc, rx.K_ldpc
    llr = (2.0 * (1.0 + 0.35 * torch.randn((F, N), device=dev, dtype=torch.float32)) / 0.35 ** 2)
    bits = torch.empty((F, K), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
    torch.cuda.synchronize()
    rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize(); ts.append(time.perf_counter() - t0)
    print("32APSK-S_3/4", F, implem, rx.ldpc_kernel_name(), "%.2f ms  %.0f k frames/s  cwd %d" % (1e3 * min(ts), F / min(ts) / 1e3, int(cwd.sum())))
    rx.close()
