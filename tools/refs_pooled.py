#!/usr/bin/env python3
"""Pooled comparison of a set of simulator runs with the reference's regression traces refs/TX_RX_BB/*.txt (tests/golden/refs_tx_rx_bb.json).

For every row (MODCOD, Eb/N0) present in both: r = FER_run / FER_ref, its log-normal one-sigma error sigma = sqrt(1 / FE_ref + 1 / FE_run) (Poisson counts: the
reference stopped at ~100 frame errors per row, i.e. +-10 % on its side alone).  Pooled over the rows with weights 1 / sigma^2:
    pooled log-ratio, its sigma, exp of both;  chi^2 of the rows against ratio 1 (and its degrees of freedom);  per trace, the weighted least-squares slope of
    log r against -log10(FER_ref) -- a drift towards the low-FER end of a trace shows as a positive slope.
A run that reproduces the reference's decoder shows a pooled ratio of 1 within ~2 sigma, chi^2 ~ dof and no slope.

usage: python tools/refs_pooled.py DIR PREFIX [--md]          files DIR/PREFIX{qpsk_8_9,qpsk_3_5,8psk_3_5,8psk_8_9,16apsk_8_9}.txt, the simulators' table format
       e.g. python tools/refs_pooled.py results/r05 ref1000_"""
import json
import math
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FILES = [("QPSK_8_9.txt", "QPSK-S_8/9", "qpsk_8_9"), ("QPSK_3_5.txt", "QPSK-S_3/5", "qpsk_3_5"), ("8PSK_3_5.txt", "8PSK-S_3/5", "8psk_3_5"),
         ("8PSK_8_9.txt", "8PSK-S_8/9", "8psk_8_9"), ("16APSK_8_9.txt", "16APSK-S_8/9", "16apsk_8_9")]


def read_run(path):
    rows = {}
    for l in open(path):
        if re.match(r"^ +[0-9]", l):
            f = [x.strip() for x in l.replace("||", "|").split("|")]
            rows[round(float(f[1]), 2)] = dict(fra=int(f[2]), be=int(f[3]), fe=int(f[4]), ber=float(f[5]), fer=float(f[6]))
    return rows


def pooled(directory, prefix):
    refs = json.load(open(os.path.join(ROOT, "tests", "golden", "refs_tx_rx_bb.json")))
    rows, traces = [], {}
    for ref_file, modcod, name in FILES:
        p = os.path.join(directory, prefix + name + ".txt")
        if not os.path.exists(p):
            continue
        run = read_run(p)
        for r in refs[ref_file]["rows"]:
            eb = round(float(r["ebn0"]), 2)
            if eb not in run or run[eb]["fe"] == 0:
                continue
            g = run[eb]
            fer_ref, fer_run = r["fe"] / r["fra"], g["fe"] / g["fra"]            # from the counts (the traces print three digits)
            lr = math.log(fer_run / fer_ref)
            sg = math.sqrt(1.0 / r["fe"] + 1.0 / g["fe"])
            row = dict(ref=ref_file, modcod=modcod, ebn0=eb, fer_ref=fer_ref, fe_ref=r["fe"], fer_run=fer_run, fe_run=g["fe"], fra_run=g["fra"], ratio=math.exp(lr), log_ratio=lr, sigma=sg,
                       z=lr / sg, x=-math.log10(fer_ref),
                       ber_log_ratio=math.log((g["be"] / g["fra"]) / (r["be"] / r["fra"])) if g["be"] and r["be"] else 0.0)
            rows.append(row)
            traces.setdefault(ref_file, []).append(row)
    if not rows:
        return None
    W = sum(1 / r["sigma"] ** 2 for r in rows)
    m = sum(r["log_ratio"] / r["sigma"] ** 2 for r in rows) / W
    s = 1 / math.sqrt(W)
    chi2 = sum((r["log_ratio"] / r["sigma"]) ** 2 for r in rows)
    mb = sum(r["ber_log_ratio"] / r["sigma"] ** 2 for r in rows) / W      # bit-error-rate ratio, pooled with the same weights (bit errors come in bursts of one frame: the frames' sigma is the honest one)
    slopes = {}
    for t, rr in traces.items():
        if len(rr) < 3:
            continue
        w = [1 / r["sigma"] ** 2 for r in rr]
        sw = sum(w)
        xb = sum(wi * r["x"] for wi, r in zip(w, rr)) / sw
        yb = sum(wi * r["log_ratio"] for wi, r in zip(w, rr)) / sw
        sxx = sum(wi * (r["x"] - xb) ** 2 for wi, r in zip(w, rr))
        b = sum(wi * (r["x"] - xb) * (r["log_ratio"] - yb) for wi, r in zip(w, rr)) / sxx
        slopes[t] = dict(slope=b, sigma=1 / math.sqrt(sxx), rows=len(rr))
    # all traces together: one common slope around each trace's own mean
    num = den = 0.0
    for t, rr in traces.items():
        w = [1 / r["sigma"] ** 2 for r in rr]
        sw = sum(w)
        xb = sum(wi * r["x"] for wi, r in zip(w, rr)) / sw
        yb = sum(wi * r["log_ratio"] for wi, r in zip(w, rr)) / sw
        num += sum(wi * (r["x"] - xb) * (r["log_ratio"] - yb) for wi, r in zip(w, rr))
        den += sum(wi * (r["x"] - xb) ** 2 for wi, r in zip(w, rr))
    return dict(rows=rows, n=len(rows), pooled_ber_ratio=math.exp(mb), pooled_log_ratio=m, pooled_sigma=s, pooled_ratio=math.exp(m), z=m / s, chi2=chi2, dof=len(rows), slopes=slopes,
                common_slope=num / den if den else None, common_slope_sigma=1 / math.sqrt(den) if den else None, rows_above_1=sum(r["ratio"] > 1 for r in rows))


def render(res, title):
    out = ["### %s" % title, "",
           "pooled FER ratio run / reference: **%.3f +- %.3f** (log-ratio %.4f +- %.4f = %.1f sigma; %d rows, %d above 1); chi^2 against ratio 1: %.1f on %d dof; "
           "common slope of log-ratio per decade of reference FER: %+.3f +- %.3f; pooled BER ratio (same weights) %.3f"
           % (res["pooled_ratio"], res["pooled_ratio"] * res["pooled_sigma"], res["pooled_log_ratio"], res["pooled_sigma"], res["z"], res["n"], res["rows_above_1"], res["chi2"], res["dof"],
              res["common_slope"] or 0.0, res["common_slope_sigma"] or 0.0, res["pooled_ber_ratio"]), "",
           "| ref file | MODCOD | Eb/N0 | ref FER (FE) | run FER (FE) | run / ref | sigma | z |", "|---|---|---|---|---|---|---|---|"]
    for r in res["rows"]:
        out.append("| %s | %s | %.2f | %.2e (%d) | %.2e (%d) | %.3f | %.3f | %+.1f |" % (r["ref"], r["modcod"], r["ebn0"], r["fer_ref"], r["fe_ref"], r["fer_run"], r["fe_run"], r["ratio"], r["sigma"], r["z"]))
    out += ["", "per-trace slope (log-ratio per decade of reference FER): " + "; ".join("%s %+.3f +- %.3f" % (t.replace(".txt", ""), v["slope"], v["sigma"]) for t, v in res["slopes"].items())]
    return "\n".join(out)


if __name__ == "__main__":
    d, pre = sys.argv[1], sys.argv[2]
    res = pooled(d, pre)
    if res is None:
        sys.exit("no rows found under %s/%s*" % (d, pre))
    if "--json" in sys.argv:
        print(json.dumps({k: v for k, v in res.items() if k != "rows"}))
    else:
        print(render(res, "%s/%s*" % (d, pre)))
