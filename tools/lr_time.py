"""L&R fine synchronizer: the one-launch form (recurrence + rotation) against DVBS2HIP_LR=unfused, device sockets, wall time per call.
GPU box only: python tools/lr_time.py"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
vp = ctypes.c_void_p
for modcod, F in (("32APSK-S_3/4", 4096), ("QPSK-N_8/9", 1024), ("QPSK-S_8/9", 4096), ("32APSK-S_3/4", 256)):
    rx = Dvbs2Hip(modcod, max_frames=F)
    n = rx.pl_frame
    torch.manual_seed(3)
    x = torch.randn((F, 2 * n), device=dev, dtype=torch.float32); y = torch.empty_like(x); y2 = torch.empty_like(x)
    FRQ = torch.empty(F, dtype=torch.float32, device=dev); PHS = torch.empty_like(FRQ); FRQ2 = torch.empty_like(FRQ)
    torch.cuda.synchronize()
    res = {}
    for rnd in range(3):
        for form in ("fused", "unfused"):
            if form == "unfused": os.environ["DVBS2HIP_LR"] = "unfused"
            else: os.environ.pop("DVBS2HIP_LR", None)
            yy, ff = (y, FRQ) if form == "fused" else (y2, FRQ2)
            fn = lambda: rx._chk(rx.L.dvbs2hip_sync_lr_synchronize_dev(rx.h, vp(x.data_ptr()), vp(ff.data_ptr()), vp(PHS.data_ptr()), vp(yy.data_ptr()), F))
            rx.sync_lr_reset(); fn(); rx.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            rx.synchronize(); res.setdefault(form, []).append((time.perf_counter() - t0) / 20 * 1e6)
    same = bool((y == y2).all().item()) and bool((FRQ == FRQ2).all().item())
    print("%-14s F %5d  %9d samples   one launch %6.1f us (%.2f TB/s of 16 B)   three kernels %6.1f us (%.2f TB/s)   identical after 21 calls each: %s" % (
        modcod, F, n * F, min(res["fused"]), 16 * n * F / min(res["fused"]) / 1e6, min(res["unfused"]), 16 * n * F / min(res["unfused"]) / 1e6, same), flush=True)
    rx.close()
