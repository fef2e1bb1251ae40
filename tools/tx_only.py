"""TX mirror (N1) of 4096 QPSK-N frames launched a few times: the target of rocprofv3 --stats runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
modcod = sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"
F = 4096
rx = Dvbs2Hip(modcod, max_frames=F)
pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device="cuda"); sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device="cuda")
sig = torch.full((F,), 0.5, dtype=torch.float32, device="cuda")
for _ in range(5): rx.tx_bb_dev(None, 1, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F)
rx.synchronize(); rx.close()
