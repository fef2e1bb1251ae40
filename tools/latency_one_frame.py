"""Kernel times (library hipEvent timers) and wall time of one fused-chain call at F = 1, 8, 64, 512 -- fixed 10 iterations -- for one MODCOD.
usage: python tools/latency_one_frame.py [modcod]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
modcod = sys.argv[1] if len(sys.argv) > 1 else "32APSK-S_3/4"
dev = torch.device("cuda", 0)
for F in (1, 8, 64, 512, 4096):
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
    sym = torch.randn((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev); got = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev)
    sig = torch.full((F,), 0.1, dtype=torch.float32, device=dev)
    f = lambda: rx.rx_bb_dev(sym.data_ptr(), sig.data_ptr(), got.data_ptr(), None, None, F)
    for _ in range(5): f()
    rx.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): f()
    rx.synchronize(); b2b = (time.perf_counter() - t0) / 50
    rx.timing_enable(True); rx.timing_reset()
    for _ in range(10): f()
    rx.synchronize()
    out = {}
    for name, kid in (("front", B.K_FRONT), ("ldpc", B.K_LDPC), ("bch", B.K_BCH)):
        ms, n = rx.timing_get(kid); out[name] = round(ms / max(n, 1) * 1e3, 1)
    rx.timing_enable(False)
    print("%s F=%d kernel %s  back-to-back %.1f us per call  kernel %s" % (modcod, F, rx.ldpc_kernel_name(), b2b * 1e6, out))
    rx.close()
