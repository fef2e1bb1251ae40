#!/usr/bin/env python3
"""gpurun_out/lv_* (tools/profile_ldpc_variants.sh) -> profiles/<tag>_ldpc_variants.md + .json: per LDPC kernel instantiation the launch
time (rocprofv3 --kernel-trace --stats), the counters per launch (separate --pmc passes) and the BOUNDED figures recomputable from them:
  vector-ALU issue = (non-transcendental SQ_INSTS_VALU x the price of the layer loop's static mix -- 2.07 cycles for the simple two-operand class, 2.6 with a literal, 4.2-4.25 everything else, tools/kernel_mix.py -- + SQ_INSTS_VALU_TRANS_F32 x 8.06; profiles/r04_probe_issue.txt
  constants table and tools/probe_dep.hip; priced at 4 / 8 until the end of round 3) / (1024 SIMDs x busy cycles)
  fabric           = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B / launch time, against 8.6 TB/s (Infinity-Cache gathers) -- FETCH_SIZE x 2 as calibrated
                     for one dword per lane (tools/calibrate_fetch.py, profiles/r02_ldpc_rocprof.md)
  algorithmic      = SURVEY 8(d): 16 B per Tanner edge and iteration + 4 (N + K) per frame, / launch time, against 8 TB/s (an EFFECTIVE rate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_mix as KM
import csv, glob, hashlib, json, os, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
sys.path.insert(0, ROOT)
from dvbs2_amd import params as P

EDGES = {"N16200_8_9.txt": 48599, "N16200_3_5.txt": 71279, "N16200_3_4.txt": 47519, "N64800_8_9.txt": 194399}

def first(pattern):
    g = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return g[-1] if g else None

rows = []
for cfg in sorted(glob.glob(os.path.join(OUT, "lv_*.cfg")), key=lambda s: int(s.split("_")[-1].split(".")[0])):
    t = os.path.basename(cfg)[:-4]
    modcod, implem, F = open(cfg).read().split(); F = int(F)
    mc = P.get_modcod(modcod)
    r = dict(modcod=modcod, implem=implem, frames=F, n_ite=10)
    ks = first(t + "_stats/*/*_kernel_stats.csv")
    if ks:
        for row in csv.DictReader(open(ks)):
            if "ldpc_wg8_kernel" in row["Name"] or "ldpc_cu1_kernel" in row["Name"]:
                r["kernel"] = row["Name"].replace("void dvbs2::", "").split("(")[0]
                r["calls"] = int(row["Calls"]); r["avg_ms"] = float(row["AverageNs"]) / 1e6; r["min_ms"] = float(row["MinNs"]) / 1e6
    pmc = {}
    for d in ("fetch", "write", "sq1", "sq2"):
        f = first("%s_%s/*/*_counter_collection.csv" % (t, d))
        if not f:
            continue
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "ldpc_wg8_kernel" in row["Kernel_Name"] or "ldpc_cu1_kernel" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                r["vgpr"] = int(row["VGPR_Count"]); r["lds"] = int(row["LDS_Block_Size"]); r["grid"] = int(row["Grid_Size"]) // int(row["Workgroup_Size"])
        for k, v in acc.items():
            pmc[k] = sum(v) / len(v)
    r["pmc"] = pmc
    if "avg_ms" in r and pmc:
        ms = r["avg_ms"]
        E = EDGES[mc.ldpc_table]
        alg = (16.0 * E * 10 + 4.0 * (mc.N_ldpc + mc.K_ldpc)) * F
        r["algorithmic_GBps"] = alg / ms / 1e6
        r["fps"] = F / ms * 1e3
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            fab = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            r["fabric_bytes"] = fab; r["fabric_GBps"] = fab / ms / 1e6; r["fabric_frac"] = r["fabric_GBps"] / 8600.0
        if "SQ_BUSY_CYCLES" in pmc and "SQ_INSTS_VALU" in pmc:
            cyc = pmc["SQ_BUSY_CYCLES"] / 32.0
            tr = pmc.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            r["busy_cycles"] = cyc
            # prices of round 4 (profiles/r04_probe_issue.txt, tools/kernel_mix.py): the non-transcendental instructions at the static mix of the kernel's layer loop
            # (2.07 the simple two-operand class, 2.6 with a literal, 4.2-4.25 everything else), the transcendentals at 8.06 cycles
            cpi, how = KM.price_kernel(r["kernel"])
            cpi = cpi or 3.97
            r["cycles_per_valu"] = cpi; r["valu_mix"] = how
            r["valu_issue_frac"] = ((pmc["SQ_INSTS_VALU"] - tr) * cpi + tr * 8.06) / (1024.0 * cyc)
            r["trans_share_of_issue"] = tr * 8.06 / ((pmc["SQ_INSTS_VALU"] - tr) * cpi + tr * 8.06)
            r["valu_per_edge_ite"] = pmc["SQ_INSTS_VALU"] * 64.0 / (F * 10.0 * E) * (360.0 / 384.0)
        if "TCC_HIT_sum" in pmc:
            r["l2_hit"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])
        if "SQ_LDS_BANK_CONFLICT" in pmc and pmc.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict_frac"] = pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"]
    rows.append(r)

sha = hashlib.sha256()
for f in ("k_ldpc_wg8.hip", "k_ldpc.hip"):
    sha.update(open(os.path.join(ROOT, "dvbs2_amd", "csrc", f), "rb").read())
out = dict(tag=tag, kernel_sha=sha.hexdigest()[:16], rows=rows)
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_ldpc_variants.json" % tag), "w"), indent=1)
L = ["# LDPC kernel instantiations under rocprofv3 (%s): the reference's own configuration (SPA, N = 16200), the short-frame NMS kernels, and the headline beside them" % tag, "",
     "Made by `tools/profile_ldpc_variants.sh` (GPU box) + `tools/summarize_ldpc_variants.py`; sources `k_ldpc_wg8.hip` + `k_ldpc.hip` sha-256 `%s`." % out["kernel_sha"],
     "10 fixed iterations, early stop off, device-resident sockets; counters are averages per launch over separate `--pmc` passes; formulas in the tool's docstring.", "",
     "| MODCOD | implem | frames | kernel | VGPR | LDS B | avg ms (min) | k frames/s | VALU issue | transcendental share of issue | VALU / edge / ite | fabric GB / launch | fabric TB/s (of 8.6) | L2 hit | LDS conflict cycles | algorithmic TB/s (of 8) |",
     "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    if "avg_ms" not in r:
        L.append("| %s | %s | %d | (no kernel-trace) |" % (r["modcod"], r["implem"], r["frames"])); continue
    g = lambda k, f="%.2f": (f % r[k]) if k in r else "-"
    L.append("| %s | %s | %d | `%s` | %s | %s | %.3f (%.3f) | %.0f | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        r["modcod"], r["implem"], r["frames"], r.get("kernel", "?"), r.get("vgpr", "-"), r.get("lds", "-"), r["avg_ms"], r["min_ms"], r["fps"] / 1e3,
        g("valu_issue_frac"), g("trans_share_of_issue"), g("valu_per_edge_ite", "%.1f"), ("%.2f" % (r["fabric_bytes"] / 1e9)) if "fabric_bytes" in r else "-",
        ("%.2f (%.2f)" % (r["fabric_GBps"] / 1e3, r["fabric_frac"])) if "fabric_GBps" in r else "-", g("l2_hit"), g("lds_conflict_frac"),
        ("%.1f (%.2f)" % (r["algorithmic_GBps"] / 1e3, r["algorithmic_GBps"] / 8000.0))))
L += ["", "## Raw counters per launch", ""]
for r in rows:
    L.append("* %s %s x%d: " % (r["modcod"], r["implem"], r["frames"]) + ", ".join("%s %.6g" % (k, v) for k, v in sorted(r["pmc"].items())))
open(os.path.join(ROOT, "profiles", "%s_ldpc_variants.md" % tag), "w").write("\n".join(L) + "\n")
print("\n".join(L[:20]))
