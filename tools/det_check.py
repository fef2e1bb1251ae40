"""Determinism of the LDPC decoders at size (GPU box): the same batch decoded twice, and its first frames decoded alone, must give the same
hard decisions, CWD and iteration counts.  python tools/det_check.py [frames [sigma [modcod ...]]]   (DET_ITE=n: iterations)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.42
n_ite = int(os.environ.get("DET_ITE", "10"))
for modcod in sys.argv[3:] or ("QPSK-N_8/9", "QPSK-S_8/9"):
    for implem in ("NMS", "SPA"):
        for es in (False, True):
            torch.manual_seed(7)
            rx = Dvbs2Hip(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=es, implem=implem)
            N, K = rx.N_ldpc, rx.K_ldpc
            llr = (2.0 * (1.0 + sigma * torch.randn((F, N), device=dev, dtype=torch.float32)) / sigma ** 2)
            outs = []
            for n in (F, F, 256):
                bits = torch.full((n, K), -1, dtype=torch.int32, device=dev); cwd = torch.full((n,), -1, dtype=torch.int8, device=dev)
                torch.cuda.synchronize()          # (the decoder runs on the handle's own stream: the fills have to be over)
                rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), n); rx.synchronize()
                outs.append((bits.clone(), cwd.clone()))
            same_twice = bool((outs[0][0] == outs[1][0]).all()) and bool((outs[0][1] == outs[1][1]).all())
            same_sub = bool((outs[0][0][:256] == outs[2][0]).all()) and bool((outs[0][1][:256] == outs[2][1]).all())
            nd = int((outs[0][0] != outs[1][0]).any(dim=1).sum())
            print(modcod, implem, "early_stop" if es else "fixed", rx.ldpc_kernel_name(), "cwd", int(outs[0][1].sum()), int(outs[1][1].sum()), "twice identical", same_twice, "(frames differing %d)" % nd, "subset identical", same_sub, flush=True)
            rx.close()
