"""one-off (round 6): device time of the two front tasks (k_agc.hip) at 4096 QPSK-S frames of samples (osf 2) -> GB/s at the boundary (8 B in + 8 B out per complex sample)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
F, n = 4096, 8370 * 2
rx = Dvbs2Hip("QPSK-S_8/9", max_frames=F)
x = torch.randn((F, 2 * n), dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
frq = torch.empty(F, dtype=torch.float32, device="cuda")
rx.sync_coarse_set_freq(0.0123)
for name, call in (("agc_kernel", lambda: rx.agc_dev(x.data_ptr(), y.data_ptr(), n, 0.5, F)),
                   ("nco_kernel", lambda: rx._chk(rx.L.dvbs2hip_sync_coarse_synchronize_dev(rx.h, x.data_ptr(), frq.data_ptr(), frq.data_ptr(), y.data_ptr(), n, F)))):
    for _ in range(3):
        call()
    rx.synchronize(); rx.timing_enable(True); rx.timing_reset()
    for _ in range(20):
        call()
    ms, k = rx.timing_get(B.K_MISC)
    rx.timing_enable(False)
    print("%s: %.3f ms per %d frames of %d complex samples = %.2f TB/s at the boundary (16 B per sample)" % (name, ms / k, F, n, 16.0 * F * n / (ms / k * 1e-3) / 1e12))
