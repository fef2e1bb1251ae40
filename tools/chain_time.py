"""One line: device time of the fused chain's three kernels and its wall time per call (QPSK-N, 4096 frames, 10 fixed iterations) -- for tools/ab_kernel.sh (AB_MODE=cmd)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B, params as P
modcod, F = (sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"), 4096
mc = P.get_modcod(modcod)
dev = torch.device("cuda", 0)
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
sigma = P.esn0_to_sigma(P.ebn0_to_esn0(4.2 if mc.bps == 2 else 9.0, mc.code_rate, mc.bps))
pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev)
sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev); got = torch.empty_like(sent)
sig = torch.full((F,), sigma, dtype=torch.float32, device=dev)
rx.tx_bb_dev(None, 1, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F); rx.synchronize()
f = lambda: rx.rx_bb_dev(pl.data_ptr(), sig.data_ptr() if mc.bps >= 4 else None, got.data_ptr(), None, None, F)
for _ in range(3): f()
rx.synchronize(); rx.timing_enable(True); rx.timing_reset()
t0 = time.perf_counter()
for _ in range(10): f()
rx.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
out = {n: rx.timing_get(k)[0] / 10 for n, k in (("front", B.K_FRONT), ("ldpc", B.K_LDPC), ("bch", B.K_BCH))}
print("chain %.3f ms  front %.3f  ldpc %.3f  bch %.3f  ok %s" % (wall, out["front"], out["ldpc"], out["bch"], bool((got == sent).all().item())))
