"""CPU oracle, paired comparison of the two sum-product check-node rules on the SAME noisy frames: ORC_SPA (exact boxplus) against ORC_SPA_TANH (the
saturating tanh-product form of AFF3CT's Update_rule_SPA), the reference's configuration otherwise (50 iterations, natural row order, syndrome early
stop, the estimator the trace's command line names).  Counts the frames each rule loses and the discordant ones (lost by one rule only): a paired
count separates the rules with far fewer frames than two independent FER estimates.  Test infrastructure: imports the oracle, never a product path.

usage: python tools/spa_rule_paired.py --mod-cod 8PSK-S_3/5 --ebn0 2.9 --frames 6000 [--workers 6] [--perfect]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def work(job):
    modcod, ebn0, perfect, seed, n, sched = job
    from oracle import oracle as O
    from helpers import chain, sigma_for
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(seed)
    sigma = sigma_for(mc, ebn0)
    fe = {"exact": 0, "tanh": 0, "exact_only": 0, "tanh_only": 0}
    be = {"exact": 0, "tanh": 0}
    ites = {"exact": 0, "tanh": 0}
    for _ in range(n):
        info = rng.integers(0, 2, mc.K_bch).astype(np.int32)
        plf, _ = ch.tx(info)
        noisy = plf + (sigma * rng.standard_normal(plf.size)).astype(np.float32)
        bad = {}
        if sched == "both":          # the tanh rule under the two sweep orders: "exact" column = natural row order (the reference's), "tanh" column = QC layers (the kernels')
            variants = (("exact", O.SPA_TANH, O.NATURAL), ("tanh", O.SPA_TANH, O.QC))
        else:
            variants = (("exact", O.SPA, O.NATURAL if sched == "natural" else O.QC), ("tanh", O.SPA_TANH, O.NATURAL if sched == "natural" else O.QC))
        for name, implem, sc in variants:
            r = ch.rx(noisy, sigma=np.float32(sigma) if perfect else None, n_ite=50, alpha=1.0, implem=implem, sched=sc, early_stop=True)
            e = int((r["info"] != info).sum())
            be[name] += e
            bad[name] = e > 0
            fe[name] += e > 0
            ites[name] += int(r["ites"])
        fe["exact_only"] += bad["exact"] and not bad["tanh"]
        fe["tanh_only"] += bad["tanh"] and not bad["exact"]
    return n, fe, be, ites


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mod-cod", default="8PSK-S_3/5")
    ap.add_argument("--ebn0", type=float, default=2.9)
    ap.add_argument("--frames", type=int, default=6000)
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--perfect", action="store_true")
    ap.add_argument("--sched", default="natural", choices=("natural", "qc", "both"), help="both: SPA_TANH under natural order (reported as `exact`) against SPA_TANH under QC layers (reported as `tanh`)")
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    per = 40
    jobs = [(a.mod_cod, a.ebn0, a.perfect, a.seed + j, per, a.sched) for j in range((a.frames + per - 1) // per)]
    t0 = time.time()
    tot = {"frames": 0, "fe": {"exact": 0, "tanh": 0, "exact_only": 0, "tanh_only": 0}, "be": {"exact": 0, "tanh": 0}, "ites": {"exact": 0, "tanh": 0}}
    with mp.Pool(a.workers) as pool:
        for n, fe, be, ites in pool.imap_unordered(work, jobs):
            tot["frames"] += n
            for k in fe:
                tot["fe"][k] += fe[k]
            for k in be:
                tot["be"][k] += be[k]
                tot["ites"][k] += ites[k]
    tot.update(modcod=a.mod_cod, ebn0=a.ebn0, sched=a.sched, perfect=a.perfect, seconds=time.time() - t0)
    d = tot["fe"]
    n_disc = d["exact_only"] + d["tanh_only"]
    # McNemar: under "the rules lose frames equally often" the discordant frames split 50 / 50
    tot["mcnemar_z"] = (d["exact_only"] - d["tanh_only"]) / max(1.0, n_disc) ** 0.5
    tot["fer_ratio_tanh_over_exact"] = d["tanh"] / d["exact"] if d["exact"] else None
    print(json.dumps(tot))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(tot, fh, indent=1)


if __name__ == "__main__":
    main()
