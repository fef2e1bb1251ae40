import sys, time, torch
sys.path.insert(0, "/root/repo")
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
dev = torch.device("cuda", 0)
cases = [("QPSK-S_8/9", int(a[1:])) if a[0] == "S" else ("QPSK-N_8/9", int(a)) for a in sys.argv[1:]] or [("QPSK-N_8/9", 4096), ("QPSK-N_8/9", 32768), ("QPSK-S_8/9", 65024)]      # usage: python tools/bench_natural.py [normal-frame counts | S<short-frame count> ..]
for modcod, F in cases:
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
    N, K = rx.N_ldpc, rx.K_ldpc
    llr = (2.0 * (1.0 + 0.35 * torch.randn((F, N), device=dev, dtype=torch.float32)) / 0.35 ** 2)
    bits = torch.empty((F, K), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
    for sched, name in ((B.SCHED_NATURAL, "natural"), (B.SCHED_QC, "qc")):
        rx.set_ldpc_schedule(sched)
        rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
        t0 = time.perf_counter()
        for _ in range(2): rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
        rx.synchronize(); dt = (time.perf_counter() - t0) / 2
        print(modcod, F, name, rx.ldpc_kernel_name(), "%.2f ms  %.0f k frames/s  cwd %d errs %d" % (dt * 1e3, F / dt / 1e3, int(cwd.sum()), int(bits.sum())), flush=True)
    rx.close(); del llr, bits
