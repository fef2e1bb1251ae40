"""Device time of the BCH decoder + BB descrambler inside the fused chain, one line (tools/ab_kernel.sh, grid sweeps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
modcod = sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, early_stop=True)
dev = torch.device("cuda", 0)
pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev); sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev); got = torch.empty_like(sent)
sig = torch.full((F,), 0.1, dtype=torch.float32, device=dev)
rx.tx_bb_dev(None, 1, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F); rx.synchronize()
f = lambda: rx.rx_bb_dev(pl.data_ptr(), None, got.data_ptr(), None, None, F)
f(); rx.synchronize(); rx.timing_enable(True); rx.timing_reset()
for _ in range(10): f()
ms, n = rx.timing_get(B.K_BCH)
print("%s bch %.4f ms" % (modcod, ms / n))
rx.close()
