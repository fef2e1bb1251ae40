"""LDPC-only timing of the SPA check node (the reference's default --dec-implem) beside NMS, fixed iterations.
GPU box only: python tools/bench_spa.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
for modcod, F in (("QPSK-S_8/9", 8192), ("QPSK-N_8/9", 4096)):
    for implem in ("NMS", "SPA"):
        rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False, implem=implem)
        N, K = rx.N_ldpc, rx.K_ldpc
        llr = (2.0 * (1.0 + 0.42 * torch.randn((F, N), device=dev, dtype=torch.float32)) / 0.42 ** 2)
        bits = torch.empty((F, K), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
        rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
        rx.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(modcod, F, implem, rx.ldpc_kernel_name(), "%.2f ms / 10 ite  %.0f k frames/s  cwd %d" % (dt * 1e3, F / dt / 1e3, int(cwd.sum())), flush=True)
        rx.close()
