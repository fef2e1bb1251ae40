"""One LDPC decoder configuration for the rocprofv3 passes of tools/profile_ldpc_variants.sh (GPU box only):
    python3 tools/pmc_ldpc_workload.py MODCOD NMS|SPA FRAMES [n_ite]
one warm-up launch + 4 launches of decode_siho on device-resident sockets, fixed iterations, LLRs of a noisy all-zero word (4 dB-ish)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
modcod, implem, F = sys.argv[1], sys.argv[2], int(sys.argv[3])
n_ite = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(1)
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=False, implem=implem)
llr = (2.0 * (1.0 + 0.42 * torch.randn((F, rx.N_ldpc), device=dev, dtype=torch.float32)) / 0.42 ** 2)
bits = torch.empty((F, rx.K_ldpc), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
for _ in range(5):
    rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
rx.synchronize()
print(modcod, implem, F, rx.ldpc_kernel_name(), "cwd", int(cwd.sum()))
rx.close()
