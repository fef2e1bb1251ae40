"""Manual bring-up script (not a test): python tests/gpu_quick.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as O
from dvbs2_amd.receiver import Dvbs2Hip
from helpers import chain, make_llrs, make_pl_frames
for modcod, eb in [("QPSK-S_8/9", 4.2), ("QPSK-S_3/5", 2.0), ("32APSK-S_3/4", 3.4), ("QPSK-N_8/9", 4.2)]:
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, 3, eb, 1)
    rx = Dvbs2Hip(modcod, max_frames=3, n_ite=10, alpha=0.875, early_stop=False)
    V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=10, alpha=0.875, sched=O.QC, early_stop=False)
    print(modcod, "bits diff", int((V != Vo).sum()), "post maxabs", float(np.abs(post - posto).max()),
          "post exact", bool(np.array_equal(post, posto)), "cwd", CWD, cwdo, "errs vs tx", int((V != cw[:, :ch.mc.K_ldpc]).sum()))
    rx.close()
