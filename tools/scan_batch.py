"""Launch time of the LDPC kernel against the number of frames in the launch (fixed iterations): the slope is the steady-state rate, the intercept what a launch pays once
(start, the last round's partly idle grid).  GPU box only: python tools/scan_batch.py [MODCOD [NMS|SPA [reps]]]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
modcod = sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"
implem = sys.argv[2] if len(sys.argv) > 2 else "NMS"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sizes = [int(x) for x in os.environ.get("SCAN_SIZES", "256 512 768 1024 1536 2048 3072 4096 4608 5120 6144 8192 16384").split()]
Fm = max(sizes)
rx = Dvbs2Hip(modcod, max_frames=Fm, n_ite=10, alpha=1.0, early_stop=False, implem=implem)
N, K = rx.N_ldpc, rx.K_ldpc
torch.manual_seed(1)
llr = (2.0 * (1.0 + 0.42 * torch.randn((Fm, N), device=dev, dtype=torch.float32)) / 0.42 ** 2)
bits = torch.empty((Fm, K), dtype=torch.int32, device=dev); cwd = torch.empty(Fm, dtype=torch.int8, device=dev)
torch.cuda.synchronize()
print(modcod, implem, rx.ldpc_kernel_name())
for F in sizes:
    rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
        rx.synchronize(); ts.append(time.perf_counter() - t0)
    print("%6d frames  %8.3f ms (min %.3f)  %7.1f k frames/s  %.3f us/frame" % (F, sum(ts) / len(ts) * 1e3, min(ts) * 1e3, F / min(ts) / 1e3, min(ts) / F * 1e6), flush=True)
rx.close()
