"""Per-call wall latency (one dvbs2hip_ldpc_decode_siho_dev + synchronize) of the N = 64800 min-sum decoder at small batches, default image (two frames per CU) against the
one-frame-per-CU image (DVBS2HIP_LDPC_FAST_MODE=cu1).  usage: python tools/latency_normal.py [implem] [n_ite]"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from dvbs2_amd.receiver import Dvbs2Hip
    implem, n_ite = sys.argv[2], int(sys.argv[3])
    dev = torch.device("cuda", 0)
    for F in (1, 8, 64, 128, 256, 512):
        rx = Dvbs2Hip("QPSK-N_8/9", max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=False, implem=implem)
        g = torch.Generator(device=dev); g.manual_seed(1)
        sg = 0.5
        x = (1.0 + sg * torch.randn((F, rx.N_ldpc), generator=g, device=dev)) * (2.0 / sg ** 2)
        c, b = torch.empty((F,), dtype=torch.int8, device=dev), torch.empty((F, rx.K_ldpc), dtype=torch.int32, device=dev)
        for _ in range(3):
            rx.decode_siho_dev(x.data_ptr(), c.data_ptr(), b.data_ptr(), F)
        rx.synchronize()
        lat = []
        for _ in range(30):
            t = time.perf_counter(); rx.decode_siho_dev(x.data_ptr(), c.data_ptr(), b.data_ptr(), F); rx.synchronize(); lat.append(time.perf_counter() - t)
        lat.sort()
        print("  F=%4d  %s  median %.3f ms  min %.3f ms  cwd %d" % (F, rx.ldpc_kernel_name(), 1e3 * lat[15], 1e3 * lat[0], int(c.sum())))
        rx.close()
else:
    implem = sys.argv[1] if len(sys.argv) > 1 else "NMS"
    n_ite = sys.argv[2] if len(sys.argv) > 2 else "10"
    for mode in ("", "cu1"):
        env = dict(os.environ)
        if mode:
            env["DVBS2HIP_LDPC_FAST_MODE"] = mode
        print("mode", mode or "default", implem, n_ite)
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", implem, n_ite], env=env)
