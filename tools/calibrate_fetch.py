"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS code's access pattern (one dword per lane,
256 B per wave-instruction), as MI355X_MICROARCH.md (HBM) asks before trusting an absolute byte count:
runs the de-interleaver as a pure copy (QPSK: identity permutation) over a buffer far larger than the
256 MiB Infinity Cache, so the known traffic is N*F*4 bytes read + N*F*4 bytes written per launch.
Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/profile_gpu.sh does)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
F = 4096
rx = Dvbs2Hip("QPSK-N_8/9", max_frames=F)
a = torch.randn((F, rx.N_ldpc), dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
for _ in range(3):
    rx.L.dvbs2hip_deinterleave_dev(rx.h, a.data_ptr(), b.data_ptr(), F)
rx.synchronize()
assert torch.equal(a, b)
print("known bytes per launch: read %d write %d" % (a.numel() * 4, a.numel() * 4))
rx.close()
