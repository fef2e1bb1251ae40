"""BASELINE configs[4] at F = 1 .. 64: the call sequence matched filter -> extraction -> fused chain, one call + synchronize at a time (bench.py _fir_config's latency rows), for a kernel trace
(tools/r05_latency_trace.sh).  usage: python tools/latency_f1.py [frames] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import params as P
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
modcod, ebn0, n_ite, osf = "32APSK-S_3/4", 14.0, 10, 2
mc = P.get_modcod(modcod); dev = torch.device("cuda", 0)
Fg = F + 1
rx = Dvbs2Hip(modcod, max_frames=Fg, n_ite=n_ite, alpha=1.0, early_stop=False)
n = rx.pl_frame
sig_sym = float(P.esn0_to_sigma(P.ebn0_to_esn0(ebn0, mc.code_rate, mc.bps)))
zero = torch.zeros((Fg,), dtype=torch.float32, device=dev)
sig_smp = torch.full((Fg,), sig_sym * 2.0 ** 0.5, dtype=torch.float32, device=dev)
sig_c = torch.full((Fg,), sig_sym, dtype=torch.float32, device=dev)
sent = torch.empty((Fg, rx.K_bch), dtype=torch.int32, device=dev); got = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev)
pl = torch.empty((Fg, 2 * n), dtype=torch.float32, device=dev)
up = torch.empty((Fg, 2 * n * osf), dtype=torch.float32, device=dev); noisy = torch.empty_like(up); mf = torch.empty_like(up)
sym = torch.zeros((Fg, 2 * n), dtype=torch.float32, device=dev)
rx.tx_bb_dev(None, 777, zero.data_ptr(), sent.data_ptr(), pl.data_ptr(), Fg)
rx.shape_filter_dev(pl.data_ptr(), up.data_ptr(), n, Fg)
rx.add_noise_dev(sig_smp.data_ptr(), up.data_ptr(), noisy.data_ptr(), 5, 2 * n * osf, Fg)
def once():
    rx.filter_reset()
    rx.filter_dev(noisy.data_ptr(), mf.data_ptr(), n * osf, Fg)
    rx.extract_dev(mf.data_ptr(), sym.data_ptr(), n, osf, 80, Fg)
    rx.rx_bb_dev(sym.data_ptr(), sig_c.data_ptr(), got.data_ptr(), None, None, F)
once(); once(); rx.synchronize()
lat = []
for _ in range(reps):
    t = time.perf_counter(); once(); rx.synchronize(); lat.append(time.perf_counter() - t)
lat.sort()
print("F=%d: latency median %.1f us, min %.1f us; frames decoded exactly %d of %d" % (F, 1e6 * lat[len(lat) // 2], 1e6 * lat[0], int((got == sent[:F]).all(dim=1).sum()), F))
rx.close()
