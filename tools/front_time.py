"""Device time of the fused front end (a7+a6+a3+a4) for one MODCOD, 4096 frames: one line (used by tools/ab_kernel.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B, params as P
modcod = sys.argv[1] if len(sys.argv) > 1 else "16APSK-N_8/9"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mc = P.get_modcod(modcod)
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=1, early_stop=True)
dev = torch.device("cuda", 0)
sigma = P.esn0_to_sigma(P.ebn0_to_esn0(8.2 if mc.bps >= 4 else 7.5 if mc.bps == 3 else 4.2, mc.code_rate, mc.bps))
pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev); sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev); got = torch.empty_like(sent)
sig = torch.full((F,), sigma, dtype=torch.float32, device=dev)
rx.tx_bb_dev(None, 1, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F); rx.synchronize()
f = lambda: rx.rx_bb_dev(pl.data_ptr(), sig.data_ptr() if mc.bps >= 4 else None, got.data_ptr(), None, None, F)
f(); rx.synchronize(); rx.timing_enable(True); rx.timing_reset()
for _ in range(10): f()
ms, n = rx.timing_get(B.K_FRONT)
alg = (8 * rx.pl_frame + 4 * rx.N_ldpc) * F
print("%s front %.4f ms  %.2f TB/s of 8 pl_frame + 4 N bytes per frame (%.2f of 8 TB/s)" % (modcod, ms / n, alg / (ms / n * 1e-3) / 1e12, alg / (ms / n * 1e-3) / 8e12))
rx.close()
