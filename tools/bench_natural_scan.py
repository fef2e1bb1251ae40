"""Natural row order: frames/s by batch size for one form (DVBS2HIP_NAT_PARTS) -- usage: python tools/bench_natural_scan.py MODCOD size [size ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import lib_binding as B
dev = torch.device("cuda", 0)
modcod = sys.argv[1]
for F in [int(x) for x in sys.argv[2:]]:
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
    llr = (2.0 * (1.0 + 0.35 * torch.randn((F, rx.N_ldpc), device=dev, dtype=torch.float32)) / 0.35 ** 2)
    bits = torch.empty((F, rx.K_ldpc), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
    rx.set_ldpc_schedule(B.SCHED_NATURAL)
    rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F); rx.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
    rx.synchronize(); dt = (time.perf_counter() - t0) / 2
    print("parts %s %s F=%d %.2f ms %.0f k frames/s cwd %d" % (os.environ.get("DVBS2HIP_NAT_PARTS", "auto"), modcod, F, dt * 1e3, F / dt / 1e3, int(cwd.sum())), flush=True)
    rx.close(); del llr, bits
