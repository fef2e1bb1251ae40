import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B, params as P
dev = torch.device("cuda", 0)
for modcod in ("QPSK-S_8/9", "32APSK-S_3/4"):
    mc = P.get_modcod(modcod); F = 1
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    sigma = P.esn0_to_sigma(P.ebn0_to_esn0(12.0, mc.code_rate, mc.bps))
    pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev); sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev)
    got = torch.empty_like(sent); sig = torch.full((F,), sigma, dtype=torch.float32, device=dev)
    rx.tx_bb_dev(None, 1, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F); rx.synchronize()
    f = lambda: rx.rx_bb_dev(pl.data_ptr(), sig.data_ptr(), got.data_ptr(), None, None, F)
    for _ in range(5): f()
    rx.synchronize()
    rx.timing_enable(True); rx.timing_reset()
    for _ in range(50): f()
    out = {}
    for name, kid in (("front", B.K_FRONT), ("ldpc", B.K_LDPC), ("bch", B.K_BCH)):
        ms, n = rx.timing_get(kid); out[name] = round(ms / max(n, 1) * 1e3, 1)
    rx.timing_enable(False)
    t0 = time.perf_counter()
    for _ in range(200): f(); rx.synchronize()
    wall = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for _ in range(200): f()
    rx.synchronize()
    thr = (time.perf_counter() - t0) / 200 * 1e6
    print(modcod, "kernel us", out, "sum", round(sum(out.values()), 1), "call+sync us %.1f" % wall, "back-to-back us %.1f" % thr)
    rx.close()
