"""Cost of the syndrome pass that the early stop adds to every iteration: 4096 normal frames of pure noise (never converge),
10 iterations with and without the early stop.  GPU box only."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
F = 4096
for es in (False, True):
    rx = Dvbs2Hip("QPSK-N_8/9", max_frames=F, n_ite=10, alpha=1.0, early_stop=es)
    llr = torch.randn((F, rx.N_ldpc), device=dev, dtype=torch.float32)      # pure noise: never converges
    bits = torch.empty((F, rx.K_ldpc), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
    rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
    rx.synchronize(); print("early_stop", es, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), "cwd", int(cwd.sum()))
    rx.close()
