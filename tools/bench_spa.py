"""LDPC-only timing of the SPA check node (the reference's default --dec-implem) beside NMS, fixed iterations.
GPU box only: python tools/bench_spa.py [frames_normal [frames_short [reps]]]
A launch is a whole number of frames per persistent workgroup: at 4096 normal frames = 8 per workgroup one straggler frame is 1/8 of the
launch (14.7 / 16.9 ms for the same SPA kernel from one process to the next), so steady-state rates are quoted on 16384-frame batches too."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
dev = torch.device("cuda", 0)
Fn = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Fs = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(1)
for modcod, F in (("QPSK-S_8/9", Fs), ("QPSK-S_3/5", Fs), ("QPSK-N_8/9", Fn)):
    if F <= 0:
        continue
    for implem in ("NMS", "SPA"):
        rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False, implem=implem)
        N, K = rx.N_ldpc, rx.K_ldpc
        llr = (2.0 * (1.0 + 0.42 * torch.randn((F, N), device=dev, dtype=torch.float32)) / 0.42 ** 2)
        bits = torch.empty((F, K), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
        torch.cuda.synchronize()          # the decoder runs on the handle's own stream: the LLRs have to be there
        rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
            rx.synchronize(); ts.append(time.perf_counter() - t0)
        dt = sum(ts) / len(ts)
        print(modcod, F, implem, rx.ldpc_kernel_name(), "%.2f ms / 10 ite (min %.2f max %.2f)  %.0f k frames/s  cwd %d" % (dt * 1e3, min(ts) * 1e3, max(ts) * 1e3, F / dt / 1e3, int(cwd.sum())), flush=True)
        rx.close()
        del llr, bits, cwd
