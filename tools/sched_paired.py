"""CPU oracle, min-sum with a fixed number of iterations on the SAME channel LLRs under three sweeps: the reference's row order (NATURAL), the QC-layer schedule of the
throughput kernels (QC: the 360 checks of a layer read one snapshot) and the QC layers' ORDER of checks processed one after the other (QC_SEQ) -- separates what the ORDER
costs from what the SNAPSHOT costs.  Test infrastructure.  usage: python tools/sched_paired.py --mod-cod QPSK-S_8/9 --ebn0 4.0 --frames 12000 --ite 10"""
import argparse, json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def work(job):
    modcod, ebn0, seed, n, ite, implem, stop = job
    from oracle import oracle as O
    from helpers import chain
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(seed)
    rate = mc.K_bch / mc.N_ldpc
    sigma = float(np.sqrt(1.0 / (2.0 * rate * 10.0 ** (ebn0 / 10.0))))
    info = rng.integers(0, 2, (n, mc.K_bch)).astype(np.int32)
    cw = ch.ldpc.encode(ch.bch.encode(info))
    llr = (2.0 * ((1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.shape)) / sigma ** 2).astype(np.float32)
    bad = {}
    for name, sc in (("natural", O.NATURAL), ("qc", O.QC), ("qc_seq", O.QC_SEQ), ("qc_fix", O.QC_FIX)):
        V, _, _, _ = ch.ldpc.decode(llr, n_ite=ite, alpha=1.0, implem=getattr(O, implem), sched=sc, early_stop=stop)
        bad[name] = (V != cw[:, :mc.K_ldpc]).any(axis=1)
    return {k: int(v.sum()) for k, v in bad.items()}, int((bad["qc"] & ~bad["natural"]).sum()), int((bad["natural"] & ~bad["qc"]).sum()), int((bad["qc_seq"] & ~bad["natural"]).sum()), int((bad["natural"] & ~bad["qc_seq"]).sum()), n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mod-cod", default="QPSK-S_8/9"); ap.add_argument("--ebn0", type=float, default=4.0); ap.add_argument("--frames", type=int, default=12000)
    ap.add_argument("--ite", type=int, default=10); ap.add_argument("--implem", default="NMS"); ap.add_argument("--workers", type=int, default=6); ap.add_argument("--out", default=""); ap.add_argument("--early-stop", action="store_true")
    a = ap.parse_args()
    per = 100
    jobs = [(a.mod_cod, a.ebn0, 500 + j, per, a.ite, a.implem, a.early_stop) for j in range((a.frames + per - 1) // per)]
    tot = {"natural": 0, "qc": 0, "qc_seq": 0, "qc_fix": 0}
    d = [0, 0, 0, 0]; N = 0
    t0 = time.time()
    with mp.Pool(a.workers) as pool:
        for fe, a1, a2, a3, a4, n in pool.imap_unordered(work, jobs):
            for k in tot: tot[k] += fe[k]
            d[0] += a1; d[1] += a2; d[2] += a3; d[3] += a4; N += n
    res = dict(modcod=a.mod_cod, ebn0=a.ebn0, ite=a.ite, implem=a.implem, frames=N, frame_errors=tot, qc_only=d[0], natural_only_vs_qc=d[1], qc_seq_only=d[2], natural_only_vs_qc_seq=d[3],
               qc_over_natural=tot["qc"] / max(1, tot["natural"]), qc_seq_over_natural=tot["qc_seq"] / max(1, tot["natural"]), qc_fix_over_natural=tot["qc_fix"] / max(1, tot["natural"]), seconds=time.time() - t0)
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
