"""Tuning sweep on the GPU box (not a test): LDPC storage policy vs throughput.
usage: python tools/gpu_sweep.py [modcod] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
modcod = sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
configs = sys.argv[3:] or ["global:-1", "global:90", "global:56", "global:37", "global:20", "global:0", "lds:-1", "lds:20", "lds:0"]
dev = torch.device("cuda", 0)
rx0 = Dvbs2Hip(modcod, max_frames=1); N, K = rx0.N_ldpc, rx0.K_ldpc; E = rx0.ldpc_edges; rx0.close()
g = torch.Generator(device=dev); g.manual_seed(1)
llr = (1.0 + 0.33 * torch.randn((F, N), generator=g, device=dev)) * (2 / 0.33 ** 2)
bits = torch.empty((F, K), dtype=torch.int32, device=dev); cwd = torch.empty((F,), dtype=torch.int8, device=dev)
ref = None
for cfg in configs:
    c2v, grp = cfg.split(":")
    os.environ["DVBS2HIP_LDPC_C2V"] = c2v
    try:
        rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False, lds_groups=int(grp))
    except Exception as e:
        print(cfg, "create failed:", e); continue
    rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
    if ref is None: ref = bits.clone()
    same = bool((bits == ref).all().item())
    rx.timing_enable(True); rx.timing_reset()
    for _ in range(3): rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
    ms, n = rx.timing_get(B.K_LDPC)
    ms /= n
    print("%-12s %8.2f ms  %9.0f frames/s  frac %.3f  same=%s nonzero_bits=%d cwd=%d" % (
        cfg, ms, F / ms * 1e3, (16 * E * 10 + 4 * N + 4 * K) * F / (ms * 1e-3) / 8e12, same, int(bits.sum().item()), int(cwd.sum().item())), flush=True)
    rx.close()
