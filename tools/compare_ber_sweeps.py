#!/usr/bin/env python3
"""Two sets of BER / FER sweeps (tools/run_ber_sweeps.sh -> ber_*.json), point by point: FRA / BE / FE equal?  The simulator's noise is counter-based, so the same
command line on the same decoder gives the same counts; a round that changes no min-sum result reproduces the previous round's tables count for count.
usage: python tools/compare_ber_sweeps.py results/r05 gpurun_out [title] > results/r06/ber_sweeps_vs_r05.md"""
import glob, json, os, sys
old, new = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else "%s against %s" % (new, old)
rows, tot = [], [0, 0, 0, 0, 0]
for p in sorted(glob.glob(os.path.join(old, "ber_*.json"))):
    q = os.path.join(new, os.path.basename(p))
    if not os.path.exists(q):
        continue
    a, b = json.load(open(p))["rows"], json.load(open(q))["rows"]
    same = sum(1 for x, y in zip(a, b) if (x["fra"], x["be"], x["fe"]) == (y["fra"], y["be"], y["fe"]))
    rows.append("| `%s` | %d | %d | %d | %d | %d of %d |" % (os.path.basename(p), len(b), sum(y["fra"] for y in b), sum(y["be"] for y in b), sum(y["fe"] for y in b), same, len(a)))
    tot[0] += len(b); tot[1] += sum(y["fra"] for y in b); tot[2] += sum(y["be"] for y in b); tot[3] += sum(y["fe"] for y in b); tot[4] += same
print("# %s\n" % title)
print("`tools/run_ber_sweeps.sh` (`python -m dvbs2_amd.sim --clones 1`: TX mirror -> AWGN -> fused RX chain -> monitor on one MI355X, counter-based noise) compared point by point "
      "with the committed tables of the earlier round (`tools/compare_ber_sweeps.py`).\n")
print("**%d points, %.1f M frames, %.1f M bit errors, %d frame errors: %d of %d points with FRA / BE / FE equal.**\n" % (tot[0], tot[1] / 1e6, tot[2] / 1e6, tot[3], tot[4], tot[0]))
print("| sweep | points | frames | bit errors | frame errors | points equal |\n|---|---|---|---|---|---|")
print("\n".join(rows))
