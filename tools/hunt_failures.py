"""Frames that fail far above the waterfall: are they the code's (the CPU oracle fails on them too) or the GPU's?  Runs the on-device Monte-Carlo
loop until a few frames fail, then replays exactly those PL frames through the oracle chain (same sigma, same decoder configuration, QC
schedule) and compares: the demapper's LLRs, the LDPC outcome, the payload.  GPU box only; test infrastructure (imports the oracle).
usage: python tools/hunt_failures.py MODCOD ESN0_DB [implem n_ite want max_batches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import params as P
from oracle import oracle as O
from helpers import chain

modcod, esn0 = sys.argv[1], float(sys.argv[2])
implem = sys.argv[3] if len(sys.argv) > 3 else "SPA"
n_ite = int(sys.argv[4]) if len(sys.argv) > 4 else 50
want = int(sys.argv[5]) if len(sys.argv) > 5 else 4
max_batches = int(sys.argv[6]) if len(sys.argv) > 6 else 200
mc = P.get_modcod(modcod)
F = 8192 if mc.N_ldpc == 16200 else 2048
dev = torch.device("cuda", 0)
rx = Dvbs2Hip(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=True, implem=implem)
sigma = P.esn0_to_sigma(esn0)
pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev)
sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev); got = torch.empty_like(sent)
sig = torch.full((F,), sigma, dtype=torch.float32, device=dev)
c0 = torch.empty(F, dtype=torch.int8, device=dev); c1 = torch.empty(F, dtype=torch.int8, device=dev)
perfect = mc.bps >= 4
ch = chain(O, modcod)
found = 0
for b in range(max_batches):
    rx.tx_bb_dev(None, (b << 8) + 7, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F)
    rx.rx_bb_dev(pl.data_ptr(), sig.data_ptr() if perfect else None, got.data_ptr(), c0.data_ptr(), c1.data_ptr(), F)
    rx.synchronize()
    bad = torch.nonzero((got != sent).any(dim=1)).flatten().tolist()
    for f in bad:
        x = pl[f].cpu().numpy(); s = sent[f].cpu().numpy(); g = got[f].cpu().numpy()
        r = ch.rx(x, sigma=np.float32(sigma) if perfect else None, n_ite=n_ite, alpha=1.0, implem=O.SPA if implem == "SPA" else O.NMS, sched=O.QC, early_stop=True)
        # the GPU's LLRs of the same frame through the stage entry points
        d = rx.pl_descramble(x[None, :]); xf = rx.remove_plh(d)
        sg = np.float32(sigma) if perfect else rx.estimate(xf)[0][0]
        llr = rx.demodulate(np.full(1, sg, np.float32), xf, deinterleave=True)[0]
        dl = np.abs(llr - r["llr"]); tol = 1e-4 * np.maximum(1.0, np.abs(r["llr"]))
        print("batch %d frame %d | GPU: bit errors %d, ldpc cwd %d, bch cwd %d | oracle: bit errors %d, ldpc cwd %d, bch cwd %d, ites %d | LLR max |diff| %.3g, beyond tolerance at %d of %d, max |LLR| %.1f"
              % (b, f, int((g != s).sum()), int(c0[f]), int(c1[f]), int((r["info"] != s).sum()), int(r["ldpc_cwd"]), int(r["bch_cwd"]), int(r["ites"]), float(dl.max()),
                 int((dl > tol).sum()), llr.size, float(np.abs(r["llr"]).max())), flush=True)
        found += 1
        if found >= want: break
    if found >= want: break
print("frames run: %d, failing frames examined: %d" % ((b + 1) * F, found))
rx.close()
