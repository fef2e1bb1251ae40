#!/usr/bin/env python3
"""profiles/<tag>_summary.md: one table of every kernel of the path from the round's measured files -- the bench line
(profiles/<tag>_bench.json), the per-kernel table (results/<tag>/kernels.json, tools/bench_kernels.py) -- with the algorithmic rate of
SURVEY 8(d) beside the rate a plain device copy reached in the same bench run.  usage: python tools/summarize_round.py r02"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
b = json.loads(open(os.path.join(ROOT, "profiles", "%s_bench.json" % tag)).read().strip().splitlines()[-1])
k = json.load(open(os.path.join(ROOT, "results", tag, "kernels.json")))
r = b["roofline"]
copy = r.get("hbm_copy_GBps_measured") or 0.0
L = ["# Round summary (%s, one MI355X; sources: `profiles/%s_bench.json`, `results/%s/kernels.json`, counters in `profiles/%s_ldpc_rocprof.md` and `profiles/%s_kernels_pmc.md`)" % (tag, tag, tag, tag, tag), "",
     "Algorithmic bytes per unit as in SURVEY.md section 8(d), against the 8000 GB/s HBM3E spec; the last column is the same rate against what a plain 1 GiB",
     "device-to-device copy reached in the bench run (%.0f GB/s read + written).  A fraction above 1 means the kernel keeps part of its state on chip." % copy, "",
     "| kernel (row) | workload | time | algorithmic GB/s | frac of 8 TB/s | of the measured copy rate |", "|---|---|---|---|---|---|"]
def row(name, wl, ms, gbs):
    L.append("| %s | %s | %.3f ms | %.0f | %.2f | %.2f |" % (name, wl, ms, gbs, gbs / 8000.0, gbs / copy if copy else 0))
row("`%s` (a1, headline)" % r["kernel"], "4096 frames N=64800 8/9, 10 iterations, `bench.py`", r["avg_launch_ms"], r.get("algorithmic_GBps", r["achieved"]))
for name, c in k.items():
    if isinstance(c, dict) and "kernels" in c:
        for kk, v in c["kernels"].items():
            if kk.startswith("bch"):      # inside the chain this kernel reads the packed hard decisions and writes only what it corrects: no 8(d) figure applies
                L.append("| %s (verify + patch; the int32 output is written by the LDPC kernel) | %s | %.3f ms | -- | -- | -- |" % (kk, name, v["ms"]))
            else:
                row(kk, name, v["ms"], v["achieved_GBps"])
        L.append("| fused chain `rx_bb_dev` | %s | %.3f ms | %.0f frames/s = %.2f Gb/s info | | |" % (name, c["chain_wall_ms"], c["chain_frames_per_s"], c["chain_info_gbps"]))
for name in ("fir_QPSK-N(66564 cplx/frame)", "fir_32APSK-S(6804 cplx/frame)"):
    for c in k.get(name, []):
        if c["frames"] >= 1024:
            row("matched filter a5 (%s)" % c["kernel"].split(" ")[0], "%d x %d complex samples" % (c["frames"], c["n_cplx"]), c["kernel_ms"], c["GBps"])
for c in k.get("upfir_N2", []):
    row("TX shaping filter N2", "%d x %d input samples" % (c["frames"], c["n_in_cplx"]), c["call_ms"], c["GBps_24B_per_input_sample"])
for name, c in k.items():
    if name.startswith("sync_N4"):
        for kk, v in c.items():
            if isinstance(v, dict):
                row(kk + " (N4)", name, v["call_ms"], v["GBps_16B_per_sample"])
bd = r.get("bounded")
L += ["", "LDPC kernel, what physically bounds it (`roofline.bounded`): fabric traffic %.1f GB per launch = %.2f TB/s = %.2f of the %.1f TB/s Infinity-Cache rate; vector issue port %.2f busy (instruction prices of profiles/r04_probe_issue.txt);"
      % (r["traffic"] / 1e9, bd["achieved"] / 1e3, bd["frac"], bd["peak"] / 1e3, bd["valu"]["frac"]) if bd else "LDPC kernel: no valid PMC traffic file for this kernel source.",
      "bytes that must cross HBM (`roofline.hbm_true`): %.2f GB per launch = %.0f GB/s = %.3f of peak." % (r["hbm_true"]["bytes_per_launch"] / 1e9, r["hbm_true"]["achieved"], r["hbm_true"]["frac"]),
      "Early stop (the reference's default), untimed for `value`: %s frames/s." % ", ".join("%.0f k at %s" % (v / 1e3, kk) for kk, v in b["extra"]["early_stop_fps"].items()),
      "", "CPU baseline beside it (`cpu_baseline`, kind `%s`): %.0f frames/s = %.0f Mb/s on %d threads of the GPU box's host (%s...)." % (
          b["cpu_baseline"]["kind"], b["cpu_baseline"]["fec_frames_per_s"], b["cpu_baseline"]["value"] / 1e6, b["cpu_baseline"]["cores"], b["cpu_baseline"]["sample"][:100])]
open(os.path.join(ROOT, "profiles", "%s_summary.md" % tag), "w").write("\n".join(L) + "\n")
print("\n".join(L))
