"""Soak of the one-launch L&R synchronizer (workgroup 0 runs the recurrence, the others wait for its words): calls issued while ANOTHER handle's persistent LDPC kernel
owns every CU, so its workgroups are placed a few at a time as LDPC workgroups retire -- the situation in which 'workgroup 0 is placed first' matters.  Every call's
output is compared with the three-kernel path's.  GPU box only: python tools/lr_soak.py [rounds]"""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
vp = ctypes.c_void_p
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Fd = 4096
dec = Dvbs2Hip("QPSK-N_8/9", max_frames=Fd, n_ite=10, alpha=1.0, early_stop=False, implem="NMS")
torch.manual_seed(1)
llr = (2.0 * (1.0 + 0.42 * torch.randn((Fd, dec.N_ldpc), device=dev, dtype=torch.float32)) / 0.42 ** 2)
bits = torch.empty((Fd, dec.K_ldpc), dtype=torch.int32, device=dev); cwd = torch.empty(Fd, dtype=torch.int8, device=dev)
bad = 0; calls = 0
for modcod, F in (("32APSK-S_3/4", 4096), ("QPSK-N_8/9", 512), ("QPSK-S_8/9", 1000)):
    rx = Dvbs2Hip(modcod, max_frames=F)
    n = rx.pl_frame
    x = torch.randn((F, 2 * n), device=dev, dtype=torch.float32)
    y = torch.empty_like(x); yref = torch.empty_like(x)
    FRQ = torch.empty(F, dtype=torch.float32, device=dev); PHS = torch.empty_like(FRQ); FRQref = torch.empty_like(FRQ)
    torch.cuda.synchronize()
    os.environ["DVBS2HIP_LR"] = "unfused"
    rx.sync_lr_reset(); rx._chk(rx.L.dvbs2hip_sync_lr_synchronize_dev(rx.h, vp(x.data_ptr()), vp(FRQref.data_ptr()), vp(PHS.data_ptr()), vp(yref.data_ptr()), F)); rx.synchronize()
    os.environ.pop("DVBS2HIP_LR")
    t0 = time.perf_counter()
    for r in range(rounds):
        dec.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), Fd)          # asynchronous, the handle's own stream: 6 ms of a full machine
        for k in range(3):
            y.zero_(); FRQ.zero_(); torch.cuda.current_stream().synchronize()          # torch's stream only (a device-wide wait would also wait for the LDPC launch)
            rx.sync_lr_reset()
            rx._chk(rx.L.dvbs2hip_sync_lr_synchronize_dev(rx.h, vp(x.data_ptr()), vp(FRQ.data_ptr()), vp(PHS.data_ptr()), vp(y.data_ptr()), F))
            rx.synchronize()
            calls += 1
            if not (bool((y == yref).all().item()) and bool((FRQ == FRQref).all().item())): bad += 1
        dec.synchronize()
    print("%-14s F %5d: %d calls under a running LDPC launch in %.1f s, %d differ from the three-kernel path" % (modcod, F, 3 * rounds, time.perf_counter() - t0, bad), flush=True)
    rx.close()
dec.close()
print("total", calls, "calls,", bad, "bad")
sys.exit(1 if bad else 0)
