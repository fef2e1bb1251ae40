import sys, time, numpy as np, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from oracle import oracle as O
from helpers import chain, make_llrs
ch = chain(O, "QPSK-N_8/9")
_, llr1, cw = make_llrs(O, "QPSK-N_8/9", 64, 4.0, seed=1)
llr = np.tile(llr1, (32, 1))
print("frames", llr.shape[0], flush=True)
for thr in (1, 16, 64, 128, 256):
    n = min(llr.shape[0], max(64, thr * 8))
    t = ch.ldpc.decode_batch_timed(llr[:n], n_ite=10, alpha=1.0, sched=O.NATURAL, threads=thr)
    t2 = ch.ldpc.decode_batch_inter_timed(llr[:n], n_ite=10, alpha=1.0, threads=thr)
    print("threads", thr, "n", n, "scalar %.0f f/s  inter16 %.0f f/s" % (n / t[-1], n / t2[-1]), flush=True)
