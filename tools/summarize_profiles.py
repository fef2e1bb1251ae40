#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs that tools/profile_gpu.sh left under gpurun_out/ into the
tracked files under profiles/ (per round): kernel-trace stats, PMC averages per launch of the
LDPC kernel, and profiles/ldpc_pmc_traffic.json (read by bench.py for roofline.traffic)."""
import csv, glob, json, os, sys, collections
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)

def first(pattern):
    g = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return g[-1] if g else None

def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 100 else name[:97] + "..."

lines = ["# rocprofv3 summary (%s) -- `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras`" % tag, ""]
ks = first("prof_stats/*/*_kernel_stats.csv")
if ks:
    lines += ["## --kernel-trace --stats (top kernels)", "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    with open(os.path.join(PROF, "%s_kernel_stats.csv" % tag), "w") as fo:
        w = csv.writer(fo)
        for i, r in enumerate(csv.reader(open(ks))):
            if i == 0:
                w.writerow(r); continue
            r[0] = short(r[0]); w.writerow(r)
            if i <= 6:
                lines.append("| `%s` | %s | %.3f | %.1f | %s |" % (r[0], r[1], float(r[2]) / 1e6, float(r[3]) / 1e3, r[4]))
    lines.append("")
pmc = {}
for d in ("prof_fetch", "prof_write", "prof_sq1", "prof_sq2"):
    f = first(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ldpc" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = (r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    for k, v in acc.items():
        pmc[k] = sum(v) / len(v)
if pmc:
    lines += ["## PMC counters, average per launch of `%s` (separate --pmc passes)" % short(meta[0]),
              "grid %s, workgroup %s, LDS %s B, VGPR %s, SGPR %s" % meta[1:], "", "| counter | value / launch |", "|---|---|"]
    for k in sorted(pmc):
        lines.append("| %s | %.6g |" % (k, pmc[k]))
    fetch_b = pmc.get("FETCH_SIZE", 0) * 1024
    write_b = pmc.get("WRITE_SIZE", 0) * 1024
    lines += ["", "FETCH_SIZE / WRITE_SIZE are in KiB: fabric-side (L2 miss) reads %.3f GB, writes %.3f GB per launch." % (fetch_b / 1e9, write_b / 1e9),
              "MI355X_MICROARCH.md (HBM): FETCH_SIZE under-reports wide coalesced streams by 2x on gfx950 and is uncalibrated for",
              "dword-per-lane accesses (this kernel's pattern), and Infinity-Cache hits are counted, so the figure is an",
              "upper bound on true HBM bytes.  traffic = FETCH_SIZE*1024 + WRITE_SIZE*1024 (raw); 2x-corrected fetch = %.3f GB." % (2 * fetch_b / 1e9)]
    # calibration on a known dword-per-lane copy (tools/calibrate_fetch.py)
    cal = {}
    for d, name in (("prof_cal_fetch", "FETCH_SIZE"), ("prof_cal_write", "WRITE_SIZE")):
        f = first(d + "/*/*_counter_collection.csv")
        if f:
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "deinterleave" in r["Kernel_Name"] and r["Counter_Name"] == name]
            if vals:
                cal[name] = sum(vals) / len(vals) * 1024
    known = 4096 * 64800 * 4
    if cal:
        lines += ["", "Calibration (known traffic %.3f GB read + %.3f GB written per launch of the dword-per-lane copy kernel, buffer >> Infinity Cache):" % (known / 1e9, known / 1e9)]
        for k, v in cal.items():
            lines.append("  %s reports %.3f GB -> correction factor %.3f" % (k, v / 1e9, known / v))
    corr_f = known / cal["FETCH_SIZE"] if "FETCH_SIZE" in cal else None
    corr_w = known / cal["WRITE_SIZE"] if "WRITE_SIZE" in cal else None
    if "TCC_HIT_sum" in pmc:
        lines.append("L2 hit rate = %.3f" % (pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])))
    # vector issue (round 4, tools/probe_issue2.hip / probe_issue3.hip, profiles/r04_probe_issue.txt): a SIMD issues a VOP1 / VOP2 / VOPC instruction every ~2.05 cycles and a
    # VOP3 / VOP3P-ENCODED one every ~4.1, whatever the number of waves on it (two waves reach the VOP2 rate, the oldest wave alone saturates the VOP3 rate), and a lone
    # wave issues one instruction of any kind per ~4.3 cycles.  Rounds 2 and 3 priced every vector instruction at 4, then at 2 cycles; the layer loop is 46 % VOP3
    # and under a tenth of it in the simple 2.07-cycle class (tools/kernel_mix.py reads the mix from the code object: probe_issue4 moved VOPC, v_cndmask_b32_e32, v_min / v_max and the
    # shifts to the 4.25 class), i.e. ~3.97 cycles per instruction.  SQ_BUSY_CYCLES is summed over the chip's 32 shader engines
    # (8 XCDs x 4): busy cycles of the launch = SQ_BUSY_CYCLES / 32; 256 CUs x 4 SIMDs.
    valu_frac = wave_issue = cyc_per_valu = vop3_share = None
    if pmc.get("SQ_INSTS_VALU") and pmc.get("SQ_BUSY_CYCLES"):
        cyc = pmc["SQ_BUSY_CYCLES"] / 32.0
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import kernel_mix as KM
            kname = meta[0].replace("void ", "").split("(")[0].replace("dvbs2::", "")
            mx = KM.mix("k_ldpc_wg8" if "wg8" in kname else "k_ldpc_cu1" if "cu1" in kname else "k_ldpc", kname)
            cyc_per_valu, vop3_share = mx[0]["cycles_per_valu"], mx[0]["vop3_share"]
        except Exception as e:
            lines.append("(tools/kernel_mix.py failed: %s; every vector instruction priced at 3.97 cycles)" % e)
        cyc_per_valu = cyc_per_valu or 3.97
        valu_frac = pmc["SQ_INSTS_VALU"] * cyc_per_valu / (1024.0 * cyc)
        insts = sum(pmc.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
        wave_issue = insts * 4.3 / (pmc.get("SQ_WAVES", 0) * cyc) if pmc.get("SQ_WAVES") else None
        lines.append("vector issue = SQ_INSTS_VALU x %.2f SIMD cycles (static mix of the layer loop, tools/kernel_mix.py: %.0f %% VOP3-encoded; 2.07 cycles for the simple two-operand class, 2.6 with a literal, 4.2-4.25 everything else) / (1024 SIMDs x %.3g busy cycles) = %.2f; "
                     "SALU: %.3g instructions on 256 scalar units = %.2f of the cycles" % (cyc_per_valu, 100 * (vop3_share or 0), cyc, valu_frac, pmc.get("SQ_INSTS_SALU", 0), pmc.get("SQ_INSTS_SALU", 0) / (256.0 * cyc)))
        if wave_issue:
            lines.append("issue slots of a wave = (VALU + SALU + LDS + VMEM instructions) x 4.3 cycles / (%d waves x busy cycles) = %.2f on average over a workgroup's waves"
                         % (pmc["SQ_WAVES"], wave_issue))
    cf, cw = (corr_f or 1.0), (corr_w or 1.0)
    lines.append("traffic (calibrated) = %.3f x FETCH + %.3f x WRITE = %.3f GB per launch" % (cf, cw, (cf * fetch_b + cw * write_b) / 1e9))
    sys.path.insert(0, ROOT)
    import subprocess
    import bench as _bench
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        head = None
    json.dump({"hbm_bytes_per_launch": cf * fetch_b + cw * write_b, "fetch_bytes_raw": fetch_b, "write_bytes_raw": write_b,
               "fetch_correction": cf, "write_correction": cw, "source": "profiles/%s_ldpc_rocprof.md" % tag,
               "kernel_sha": _bench.kernel_sha(), "kernel_sources": list(_bench.KERNEL_SOURCES), "git_head": head,
               "kernel": meta[0].replace("void ", "").split("(")[0].replace("dvbs2::", "").replace(", false>", ">").replace(", 0>", ">").replace(", ", ","),
               "frames": _bench.FRAMES_PER_GPU, "n_ite": _bench.N_ITE,
               "valu_occupancy": valu_frac, "valu_cycles_per_inst": cyc_per_valu, "vop3_share": vop3_share, "wave_issue_occupancy": wave_issue, "valu_insts_per_launch": pmc.get("SQ_INSTS_VALU"), "busy_cycles_per_launch": (pmc.get("SQ_BUSY_CYCLES") or 0) / 32.0,
               "l2_hit_rate": (pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])) if "TCC_HIT_sum" in pmc else None,
               "note": "fabric-side bytes (L2 misses + written-through stores), Infinity-Cache hits included; corrected by the factors "
                       "measured on a known dword-per-lane copy (tools/calibrate_fetch.py)"},
              open(os.path.join(PROF, "ldpc_pmc_traffic.json"), "w"), indent=1)
open(os.path.join(PROF, "%s_ldpc_rocprof.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
