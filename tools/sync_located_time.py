"""Chained device form of the frame synchronizer (dvbs2hip_sync_frame_locate_dev -> dvbs2hip_rx_bb_located_dev) against the delayed copy
(dvbs2hip_sync_frame_synchronize_dev -> dvbs2hip_rx_bb_dev): wall time per call of the synchronizer part and of synchronizer + fused chain, a locked stream.
usage: python tools/sync_located_time.py [modcod] [frames] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import params as P
modcod = sys.argv[1] if len(sys.argv) > 1 else "32APSK-S_3/4"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mc = P.get_modcod(modcod)
dev = torch.device("cuda", 0)
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
n, K = rx.pl_frame, rx.K_bch
ebn0 = 14.0 if mc.bps >= 4 else 7.0
sig = torch.full((F,), P.esn0_to_sigma(P.ebn0_to_esn0(ebn0, mc.code_rate, mc.bps)), dtype=torch.float32, device=dev)
pl = torch.empty((F + 1, 2 * n), dtype=torch.float32, device=dev)
sent = torch.empty((F + 1, K), dtype=torch.int32, device=dev)
rx2 = Dvbs2Hip(modcod, max_frames=F + 1, n_ite=10, alpha=1.0, early_stop=False)
rx2.tx_bb_dev(None, 7, torch.full((F + 1,), float(sig[0]), dtype=torch.float32, device=dev).data_ptr(), sent.data_ptr(), pl.data_ptr(), F + 1); rx2.synchronize(); rx2.close()
off = 1234
x = pl.reshape(-1)[2 * (n - off):2 * (n - off) + F * 2 * n].contiguous()      # a stream that starts `off` symbols before a frame start
DEL = torch.empty(F, dtype=torch.int32, device=dev); FLG = torch.empty_like(DEL); TRI = torch.empty(F, dtype=torch.float32, device=dev)
Y = torch.empty_like(x); SRC = torch.zeros(F, dtype=torch.int64, device=dev)
bits = torch.empty((F, K), dtype=torch.int32, device=dev)
sg = sig.data_ptr() if mc.bps >= 4 else None
def t(fn):
    for _ in range(3): fn()
    rx.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    rx.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
res = {}
res["sync copy"] = t(lambda: rx.sync_frame_synchronize_dev(x.data_ptr(), DEL.data_ptr(), FLG.data_ptr(), TRI.data_ptr(), Y.data_ptr(), F))
res["sync located"] = t(lambda: rx.sync_frame_locate_dev(x.data_ptr(), DEL.data_ptr(), FLG.data_ptr(), TRI.data_ptr(), SRC.data_ptr(), F))
def a():
    rx.sync_frame_synchronize_dev(x.data_ptr(), DEL.data_ptr(), FLG.data_ptr(), TRI.data_ptr(), Y.data_ptr(), F); rx.rx_bb_dev(Y.data_ptr(), sg, bits.data_ptr(), None, None, F)
def b():
    rx.sync_frame_locate_dev(x.data_ptr(), DEL.data_ptr(), FLG.data_ptr(), TRI.data_ptr(), SRC.data_ptr(), F); rx.rx_bb_located_dev(SRC.data_ptr(), sg, bits.data_ptr(), None, None, F)
res["sync + chain, copy"] = t(a)
res["sync + chain, located"] = t(b)
rx.timing_enable(True); rx.timing_reset(); b(); rx.synchronize()
from dvbs2_amd import lib_binding as B
fr = rx.timing_get(B.K_FRONT)[0]
nbytes = 16.0 * n * F
print("%s F=%d locked delay %d flag %d: " % (modcod, F, int(DEL[-1]), int(FLG[-1])) + "  ".join("%s %.4f ms" % kv for kv in res.items())
      + "  | sync part of the located form = %.2f of HBM (16 B per sample over 8 TB/s); front end reading in place %.4f ms" % (nbytes / (res["sync located"] * 1e-3) / 8e12, fr))
rx.close()
