"""SPA kernel against the oracle's boxplus recursions over the ranges the complement-product form has to survive (GPU box only):
posterior error relative to max(1, |L|) after n fixed iterations, from the waterfall up to LLRs of several hundred."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dvbs2_amd.receiver import Dvbs2Hip
from oracle import oracle as O
from helpers import chain, make_llrs

for modcod, ebn0, n_ite, F, scale in (("QPSK-S_8/9", 3.9, 1, 3, 1.0), ("QPSK-S_8/9", 3.9, 2, 3, 1.0), ("QPSK-S_8/9", 5.0, 10, 3, 1.0), ("QPSK-S_8/9", 9.0, 10, 3, 1.0),
                                     ("QPSK-S_8/9", 9.0, 20, 2, 4.0), ("QPSK-S_3/5", 6.0, 10, 2, 1.0), ("32APSK-S_3/4", 8.0, 10, 2, 1.0), ("QPSK-N_8/9", 3.9, 2, 2, 1.0),
                                     ("QPSK-N_8/9", 6.0, 10, 2, 1.0), ("QPSK-N_8/9", 9.0, 10, 1, 3.0)):
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=23)
    llr = (llr * scale).astype(np.float32)
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=n_ite, early_stop=False, implem="SPA")
    V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=n_ite, implem=O.SPA, sched=O.QC, early_stop=False)
    rel = np.abs(post - posto) / np.maximum(1.0, np.abs(posto))
    print("%-13s %4.1f dB x%.0f  %2d ite  max|L| %8.1f  max rel err %.2e  nan %d  hard diff %d  cwd %s/%s  %s" % (
        modcod, ebn0, scale, n_ite, float(np.abs(posto).max()), float(np.nanmax(rel)), int(np.isnan(post).sum()), int((V != Vo).sum()), CWD.tolist(), cwdo.tolist(), rx.ldpc_kernel_name()), flush=True)
    rx.close()
