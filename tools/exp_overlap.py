"""Experiment: does a streaming kernel on a LOW-priority stream hide in the tail of the persistent LDPC kernel on a high-priority one?
Two handles on one device, each on its own torch stream (cfg.stream).  Prints per-batch times: LDPC alone, front alone, back to back on
one stream, and concurrently on two streams."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import params as P
dev = torch.device("cuda", 0)
modcod, F = "QPSK-N_8/9", 4096
lo, hi = -1, 0
try:
    s_hi, s_lo = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)      # torch: lower number = higher priority
except Exception as e:
    print("no priorities", e); s_hi, s_lo = torch.cuda.Stream(), torch.cuda.Stream()
A = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False, stream=s_hi.cuda_stream)
Bh = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False, stream=s_lo.cuda_stream)
mc = P.get_modcod(modcod)
llr = torch.randn((F, A.N_ldpc), dtype=torch.float32, device=dev) * 4 + 6
bits = torch.empty((F, A.K_ldpc), dtype=torch.int32, device=dev); cwd = torch.empty(F, dtype=torch.int8, device=dev)
sym = torch.randn((F, 2 * A.N_xfec), dtype=torch.float32, device=dev); sig = torch.full((F,), 0.4, dtype=torch.float32, device=dev)
llr2 = torch.empty((F, A.N_ldpc), dtype=torch.float32, device=dev)
vp = ctypes.c_void_p
ldpc = lambda h: h.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), F)
front = lambda h: h._chk(h.L.dvbs2hip_demodulate_deinterleave_dev(h.h, vp(sig.data_ptr()), vp(sym.data_ptr()), vp(llr2.data_ptr()), F))
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
print("ldpc alone      %.3f ms" % timeit(lambda: ldpc(A)))
print("front alone     %.3f ms" % timeit(lambda: front(A)))
print("same stream     %.3f ms" % timeit(lambda: (front(A), ldpc(A))))
print("two streams     %.3f ms (front on the low-priority stream, enqueued first)" % timeit(lambda: (front(Bh), ldpc(A))))
print("two streams     %.3f ms (ldpc enqueued first)" % timeit(lambda: (ldpc(A), front(Bh))))
A.close(); Bh.close()
