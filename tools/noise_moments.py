"""Moments of the on-device AWGN (k_tx.hip awgn_kernel / the TX mirror's noise: Philox4x32-10 + Box-Muller on the hardware log / sqrt / sin / cos units) over ~2^31 samples:
mean, variance, kurtosis, the tail beyond 4 / 5 sigma against the normal law.  A FER offset of 5 % against the reference on the refs' slopes is 0.002 dB = 0.05 % of noise
power: the variance is measured here to 0.003 % (one sigma).  usage: python tools/noise_moments.py"""
import math
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dvbs2_amd.receiver import Dvbs2Hip

F, n = 4096, 2 * 8370            # floats per frame (QPSK-S PL frame)
rx = Dvbs2Hip("QPSK-S_8/9", max_frames=F, n_ite=1, early_stop=False)
dev = torch.device("cuda", 0)
x = torch.zeros((F, n), dtype=torch.float32, device=dev)
y = torch.empty_like(x)
sig = torch.ones((F,), dtype=torch.float32, device=dev)
s1 = s2 = s4 = 0.0
t4 = t5 = 0
N = 0
for k in range(32):
    rx.add_noise_dev(sig.data_ptr(), x.data_ptr(), y.data_ptr(), 1000 + k, n, F)
    rx.synchronize()
    d = y.double()
    s1 += float(d.sum()); s2 += float((d * d).sum()); s4 += float((d ** 4).sum())
    t4 += int((y.abs() > 4).sum()); t5 += int((y.abs() > 5).sum())
    N += d.numel()
m, v = s1 / N, s2 / N
print("samples %d  mean %+.2e (sigma %.1e)  variance %.6f (sigma %.1e)  kurtosis %.5f (sigma %.1e)" % (N, m, 1 / math.sqrt(N), v, math.sqrt(2.0 / N), s4 / N / v ** 2, math.sqrt(96.0 / N)))
print("P(|n| > 4) %.4e (normal %.4e)   P(|n| > 5) %.4e (normal %.4e)" % (t4 / N, math.erfc(4 / math.sqrt(2)), t5 / N, math.erfc(5 / math.sqrt(2))))
