"""GPU box: time of the a9 monitor kernels on a BASELINE-size batch (tools/time_monitor.py [modcod] [frames])."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dvbs2_amd.receiver import Dvbs2Hip
modcod = sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rx = Dvbs2Hip(modcod, max_frames=F)
dev = torch.device("cuda", 0)
U = torch.randint(0, 2, (F, rx.K_bch), dtype=torch.int32, device=dev)
V = U.clone(); V[::7, 5] ^= 1
vp = ctypes.c_void_p
for name in ("dvbs2hip_monitor_check_errors_dev",):
    fn = getattr(rx.L, name)
    for _ in range(3): rx._chk(fn(rx.h, vp(U.data_ptr()), vp(V.data_ptr()), F))
    rx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rx._chk(fn(rx.h, vp(U.data_ptr()), vp(V.data_ptr()), F))
    rx.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s %s F=%d: %.3f ms = %.2f TB/s of the two int32 sockets" % (name, modcod, F, dt * 1e3, 2 * F * rx.K_bch * 4 / dt / 1e12))
print(rx.monitor_get())
