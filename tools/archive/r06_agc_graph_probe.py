"""one-off (round 6): the C++ RX graph with the two gain stages on the 16APSK case of tests/test_host_cpp.py: which output frames are wrong?"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from dvbs2_amd import params as P
from helpers import make_pl_frames
modcod, F = sys.argv[1], int(sys.argv[2])
n_fr, off = 6 * F if F > 1 else 10, 1777
info, pl, _, _ = make_pl_frames(O, modcod, n_fr, 14.0, seed=63)
n = pl.shape[1] // 2
stream = np.concatenate([np.zeros(2 * off, np.float32), pl.reshape(-1)])[:n_fr * 2 * n]
shaped = O.upfir(P.rrc_taps(0.2, 2, 20), 2, np.zeros(2 * 80, np.float32), stream)
d = tempfile.mkdtemp()
pin, psrc, pout = (os.path.join(d, x) for x in ("rx.f32", "src.i32", "out.i32"))
shaped.astype(np.float32).tofile(pin); info.astype(np.int32).tofile(psrc)
r = subprocess.run([os.path.join(ROOT, "host", "dvbs2_rx_bb"), "--matched-filter", "--mod-cod", modcod, "-F", str(F), "--dec-implem", "NMS", "--dec-ite", "10", "--in", pin, "--src", psrc,
                    "--src-delay", "1", "--mon-skip", sys.argv[3], "--out", pout], capture_output=True, text=True)
print(r.stdout, r.stderr[-300:])
out = np.fromfile(pout, dtype=np.int32).reshape(-1, info.shape[1])
for f in range(out.shape[0]):
    e = [(int((out[f] != info[k]).sum()), k) for k in range(n_fr)]
    print("output frame", f, "closest payload", min(e))
