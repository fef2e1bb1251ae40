"""Fused RX chain at the bench's size, once per call: wall time per call and the LDPC kernel's own time inside it (hipEvents), next to the LDPC bits socket alone on
the same box.  DVBS2HIP_LIB selects another build (tools/build_variant_tus.sh).   usage: python tools/chain_time.py [modcod] [frames] [n_ite]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dvbs2_amd import lib_binding as B
from dvbs2_amd.receiver import Dvbs2Hip

modcod = sys.argv[1] if len(sys.argv) > 1 else "QPSK-N_8/9"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_ite = int(sys.argv[3]) if len(sys.argv) > 3 else 10
r = bench._chain_config(Dvbs2Hip, torch, B, modcod, n_ite, 4.0 if "QPSK" in modcod else 8.2, F, torch.device("cuda", 0), 0, 0, reps=7)
print("chain %s F=%d: %.3f ms per call, LDPC kernel inside it %.3f ms, rest %.3f ms (%.1f %%), bit errors %d"
      % (modcod, F, r["ms"], r["ldpc_kernel_ms"], r["ms"] - r["ldpc_kernel_ms"], 100 * (r["ms"] - r["ldpc_kernel_ms"]) / r["ms"], r["bit_errors"]))
