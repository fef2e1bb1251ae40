import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
dev = torch.device("cuda", 0)
F = 4096
ref = None
for modcod in (sys.argv[1],):
    ref = None
    rx0 = Dvbs2Hip(modcod, max_frames=1); N, K, E = rx0.N_ldpc, rx0.K_ldpc, rx0.ldpc_edges; rx0.close()
    g = torch.Generator(device=dev); g.manual_seed(1)
    llr = (1.0 + 0.33 * torch.randn((F, N), generator=g, device=dev)) * (2 / 0.33 ** 2)
    bits = torch.empty((F, K), dtype=torch.int32, device=dev); cwd = torch.empty((F,), dtype=torch.int8, device=dev)
    for cfg in sys.argv[2:]:
        nf, bpc = cfg.split(":")
        os.environ["DVBS2HIP_LDPC_FAST_MODE"] = nf; os.environ["DVBS2HIP_LDPC_WF"] = bpc
        rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
        rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
        if ref is None: ref = bits.clone()
        rx.timing_enable(True); rx.timing_reset()
        for _ in range(5): rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
        ms, n = rx.timing_get(B.K_LDPC); ms /= n
        print(modcod + " MODE=%s WF=%s %8.2f ms %9.0f frames/s frac %.3f same=%s cwd=%d" % (nf, bpc, ms, F / ms * 1e3, (16 * E * 10 + 4 * N + 4 * K) * F / (ms * 1e-3) / 8e12,
              bool((bits == ref).all().item()), int(cwd.sum().item())), flush=True)
        rx.close()
