"""BASELINE config 1 ("plumbing, no GPU"): the dvbs2_tx_rx_bb chain on the CPU with the ORACLE modules
wired like /root/reference src/mains/TX_RX_BB/main.cpp:75-94 -- QPSK-S_8/9, K = 14232, -F 8,
--dec-implem NMS --dec-ite 10 (and SPA-50 for the refs comparison).  Test infrastructure: it imports
the oracle and must never be used as a product path.  usage: python tools/cpu_tx_rx_bb.py [max_frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
from dvbs2_amd import params as P
from helpers import chain, sigma_for

def run(modcod, implem, n_ite, sched, ebn0s, F=8, max_fe=100, max_frames=400, seed=1):
    ch = chain(O, modcod); mc = ch.mc
    rng = np.random.default_rng(seed)
    name = {O.NMS: "NMS", O.SPA: "SPA"}[implem]
    print("# %s  %s %d ite  schedule %s  -F %d (CPU oracle)" % (modcod, name, n_ite, "natural" if sched == O.NATURAL else "QC-layer", F))
    print("#     Es/N0 |    Eb/N0 ||      FRA |       BE |       FE |      BER |      FER ||  SIM_THR(Mb/s)")
    for ebn0 in ebn0s:
        sigma = sigma_for(mc, ebn0); esn0 = P.ebn0_to_esn0(ebn0, mc.code_rate, mc.bps)
        fra = be = fe = 0; t0 = time.time()
        while fe < max_fe and fra < max_frames:
            for _ in range(F):                                    # one "-F 8" socket worth of frames
                info = rng.integers(0, 2, mc.K_bch).astype(np.int32)
                plf, _ = ch.tx(info)
                noisy = plf + (sigma * rng.standard_normal(plf.size)).astype(np.float32)
                r = ch.rx(noisy, n_ite=n_ite, alpha=1.0, implem=implem, sched=sched, early_stop=True)
                e = int((r["info"] != info).sum()); fra += 1; be += e; fe += e > 0
        et = time.time() - t0
        print("  %9.2f | %8.2f || %8d | %8d | %8d | %8.2e | %8.2e || %8.3f" % (esn0, ebn0, fra, be, fe, be / (fra * mc.K_bch), fe / fra, fra * mc.K_bch / et / 1e6), flush=True)

mf = int(sys.argv[1]) if len(sys.argv) > 1 else 400
run("QPSK-S_8/9", O.NMS, 10, O.NATURAL, [3.6, 3.8, 4.0], max_frames=mf)
run("QPSK-S_8/9", O.NMS, 10, O.QC, [3.6, 3.8, 4.0], max_frames=mf)
run("QPSK-S_8/9", O.SPA, 50, O.NATURAL, [3.6, 3.7], max_frames=mf)
