"""One FIR workload (1024 QPSK-N frames = 68 M complex samples) launched a few times: the target of rocprofv3 runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
n_cplx, F = 66564, 1024
rx = Dvbs2Hip("32APSK-S_3/4", max_frames=F)
x = torch.randn((F, 2 * n_cplx), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
for _ in range(5): rx.filter_dev(x.data_ptr(), y.data_ptr(), n_cplx, F)
rx.synchronize(); rx.close()
