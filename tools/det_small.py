import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch, numpy as np
    from dvbs2_amd.receiver import Dvbs2Hip
    implem = sys.argv[1]
    dev = torch.device("cuda", 0)
    for F in (64, 256, 300, 512, 700, 1100):
        rx = Dvbs2Hip("QPSK-N_8/9", max_frames=F, n_ite=10, alpha=1.0, early_stop=False, implem=implem)
        g = torch.Generator(device=dev); g.manual_seed(1)
        sg = 0.5
        x = (1.0 + sg * torch.randn((F, rx.N_ldpc), generator=g, device=dev)) * (2.0 / sg ** 2)
        torch.cuda.synchronize()
        outs = []
        for k in range(3):
            c, b = torch.zeros((F,), dtype=torch.int8, device=dev), torch.zeros((F, rx.K_ldpc), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            rx.decode_siho_dev(x.data_ptr(), c.data_ptr(), b.data_ptr(), F); rx.synchronize()
            outs.append((c.cpu().numpy().copy(), b.cpu().numpy().copy()))
        same = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:])
        import hashlib
        print("  %s F=%4d %s cwd %s repeatable %s hash %s" % (implem, F, rx.ldpc_kernel_name(), [int(o[0].sum()) for o in outs], same, hashlib.md5(outs[0][1].tobytes()).hexdigest()[:8]))
        rx.close()
else:
    for mode in ("", "cu1"):
        for implem in ("NMS", "SPA"):
            env = dict(os.environ)
            if mode: env["DVBS2HIP_LDPC_FAST_MODE"] = mode
            print("mode", mode or "default"); sys.stdout.flush()
            subprocess.run([sys.executable, __file__, implem], env=env)
