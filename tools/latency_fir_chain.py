"""Where the wall latency of one F-frame call sequence of BASELINE config [4] goes (32APSK-S_3/4: filter_reset -> matched filter -> extraction -> fused chain, then synchronize):
each prefix of the sequence timed by itself.   usage: python tools/latency_fir_chain.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import params as P
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
modcod, osf = "32APSK-S_3/4", 2
dev = torch.device("cuda", 0)
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
n = rx.pl_frame
noisy = torch.randn((F, 2 * n * osf), dtype=torch.float32, device=dev); mf = torch.empty_like(noisy)
sym = torch.zeros((F, 2 * n), dtype=torch.float32, device=dev); got = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev)
sig = torch.full((F,), 0.1, dtype=torch.float32, device=dev)
steps = [("filter_reset", lambda: rx.filter_reset()), ("filter_dev", lambda: rx.filter_dev(noisy.data_ptr(), mf.data_ptr(), n * osf, F)),
         ("extract_dev", lambda: rx.extract_dev(mf.data_ptr(), sym.data_ptr(), n, osf, 80, F)), ("rx_bb_dev", lambda: rx.rx_bb_dev(sym.data_ptr(), sig.data_ptr(), got.data_ptr(), None, None, F))]
def run(k):
    for _, f in steps[:k]: f()
    rx.synchronize()
for k in range(len(steps) + 1):
    for _ in range(5): run(k)
    lat = []
    for _ in range(50):
        t = time.perf_counter(); run(k); lat.append(time.perf_counter() - t)
    lat.sort()
    print("F=%d  %-60s median %7.1f us  min %7.1f us" % (F, " + ".join(s for s, _ in steps[:k]) or "(synchronize alone)", 1e6 * lat[25], 1e6 * lat[0]))
f = steps[3][1]
for _ in range(5): f(); rx.synchronize()
lat = []
for _ in range(50):
    t = time.perf_counter(); f(); rx.synchronize(); lat.append(time.perf_counter() - t)
lat.sort(); print("F=%d  rx_bb_dev alone                                              median %7.1f us  min %7.1f us" % (F, 1e6 * lat[25], 1e6 * lat[0]))
rx.close()
