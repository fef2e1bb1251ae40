"""Device time of the TX shaping filter (row N2, fir_mfma_kernel<2>) at the two sizes of tools/bench_kernels.py: one line each (A/B tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
dev = torch.device("cuda", 0)
for n_in, F in ((3402, 4096), (33282, 1024)):
    rx = Dvbs2Hip("32APSK-S_3/4", max_frames=F)
    x = torch.randn((F, 2 * n_in), dtype=torch.float32, device=dev); y = torch.empty((F, 4 * n_in), dtype=torch.float32, device=dev)
    import time
    f = lambda: rx.shape_filter_dev(x.data_ptr(), y.data_ptr(), n_in, F)
    f(); rx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    rx.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print("upfir n_in %d x %d frames: %.4f ms  %.2f TB/s of 24 B per input sample (%.2f of 8 TB/s)" % (n_in, F, ms, 24.0 * n_in * F / (ms * 1e-3) / 1e12, 24.0 * n_in * F / (ms * 1e-3) / 8e12))
    rx.close()
