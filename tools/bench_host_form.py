"""PCIe-inclusive rate of the host-pointer (socket-semantics) entry points: H2D copy + kernels + D2H copy per call,
numpy arrays in pageable host memory.  Never the bench.py `value` (that one has its inputs resident in HBM).
GPU box only: python tools/bench_host_form.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dvbs2_amd.receiver import Dvbs2Hip
for modcod, F in (("QPSK-N_8/9", 4096), ("QPSK-S_8/9", 8192)):
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
    rng = np.random.default_rng(1)
    llr = (2.0 * (1.0 + 0.4 * rng.standard_normal((F, rx.N_ldpc), dtype=np.float32)) / 0.16).astype(np.float32)
    rx.decode_siho(llr)
    t0 = time.perf_counter(); rx.decode_siho(llr); dt = time.perf_counter() - t0
    gb = F * (rx.N_ldpc + rx.K_ldpc) * 4 / 1e9
    print("%s F=%d  ldpc_decode_siho (host sockets): %.1f ms = %.0f k frames/s, %.2f GB over PCIe (%.1f GB/s)" % (modcod, F, dt * 1e3, F / dt / 1e3, gb, gb / dt), flush=True)
    pl = rng.standard_normal((F, 2 * rx.pl_frame), dtype=np.float32)
    rx.rx_bb(pl, sigma=np.float32(0.3))
    t0 = time.perf_counter(); rx.rx_bb(pl, sigma=np.float32(0.3)); dt = time.perf_counter() - t0
    gb = F * (2 * rx.pl_frame + rx.K_bch) * 4 / 1e9
    print("%s F=%d  rx_bb (host sockets): %.1f ms = %.0f k frames/s, %.2f GB over PCIe (%.1f GB/s)" % (modcod, F, dt * 1e3, F / dt / 1e3, gb, gb / dt), flush=True)
    # the same with the sockets pinned once (dvbs2hip_host_register): chunked, copies and kernels overlapped
    V, CWD = np.empty((F, rx.K_ldpc), np.int32), np.zeros(F, np.int8)
    info, c0, c1 = np.empty((F, rx.K_bch), np.int32), np.zeros(F, np.int8), np.zeros(F, np.int8)
    for a in (llr, V, CWD, pl, info, c0, c1): rx.host_register(a)
    Vref, _ = rx.decode_siho(llr.copy())
    rx.decode_siho(llr, out=(V, CWD))
    assert np.array_equal(V, Vref)
    t0 = time.perf_counter(); rx.decode_siho(llr, out=(V, CWD)); dt = time.perf_counter() - t0
    gb = F * (rx.N_ldpc + rx.K_ldpc) * 4 / 1e9
    print("%s F=%d  ldpc_decode_siho (PINNED host sockets): %.1f ms = %.0f k frames/s (%.1f GB/s)" % (modcod, F, dt * 1e3, F / dt / 1e3, gb / dt), flush=True)
    rx.rx_bb(pl, sigma=np.float32(0.3), out=(info, c0, c1))
    t0 = time.perf_counter(); rx.rx_bb(pl, sigma=np.float32(0.3), out=(info, c0, c1)); dt = time.perf_counter() - t0
    gb = F * (2 * rx.pl_frame + rx.K_bch) * 4 / 1e9
    print("%s F=%d  rx_bb (PINNED host sockets): %.1f ms = %.0f k frames/s (%.1f GB/s)" % (modcod, F, dt * 1e3, F / dt / 1e3, gb / dt), flush=True)
    rx.close()
