#!/usr/bin/env python3
"""README.md's table "the numbers that matter", GENERATED from the round's bench line (profiles/<tag>_bench.json = bench.py's JSON line on one MI355X) and the pooled
comparison with the reference's traces (results/<tag>/r06_*.txt through tools/refs_pooled.py).  usage: python tools/make_readme_table.py r06 [--write] [--bench path]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
BEGIN, END = "<!-- BEGIN GENERATED HEADLINE TABLE (tools/make_readme_table.py) -->", "<!-- END GENERATED HEADLINE TABLE -->"


def floor_row(rd):
    """the error floor of the uncapped rule, from results/<tag>/floor_spa.txt (tools/r06_floor_spa.sh)"""
    import re
    p = os.path.join(rd, "floor_spa.txt")
    if not os.path.exists(p):
        return "| error floor | (results missing) | | |"
    cur, t = None, {}
    for l in open(p):
        m = re.match(r"^== --mod-cod (\S+) .*--dec-implem (\S+) ", l)
        if m:
            cur = (m.group(1), m.group(2)); continue
        m = re.match(r"^ +(-?[0-9.]+) \| +([0-9.]+) \|\| +(\d+) \| +(\d+) \| +(\d+) \|", l)
        if m and cur:
            t.setdefault((cur[0], float(m.group(2))), {})[cur[1]] = (int(m.group(3)), int(m.group(5)))
    k = ("QPSK-S_3/5", 1.7)
    a, b = t[k]["SPA"], t[k]["SPA_EXACT"]
    k2 = ("QPSK-S_8/9", 4.2)
    c, d = t[k2]["SPA"], t[k2]["SPA_EXACT"]
    return ("| no error floor: `SPA` (AFF3CT's message cap) against rounds 1-5's uncapped rule, same %.0f M frames | QPSK-S 3/5 at 1.7 dB: **%d** frame errors against %d; QPSK-S 8/9 at 4.2 dB: %d against %d "
            "| FER %.1e against %.1e | `test_uncapped_sum_product_has_an_error_floor_the_default_does_not` |" % (a[0] / 1e6, a[1], b[1], c[1], d[1], a[1] / a[0], b[1] / b[0]))


def loop_row(rd):
    """the filtered loop and the synchronizers in it (results/<tag>/filtered_loop_*.json, sync_in_loop.json) against the reference's full-chain traces"""
    try:
        g = json.load(open(os.path.join(rd, "filtered_loop_genie.json")))["rows"]
        b = json.load(open(os.path.join(rd, "filtered_loop_bb.json")))["rows"]
        s = json.load(open(os.path.join(rd, "sync_in_loop.json")))["rows"]
    except OSError:
        return "| the filtered loop (shaping filter .. matched filter) against the reference's full-chain traces | (results missing) | | |"
    rat = [x["fer"] / y["fer"] for x, y in zip(g, b)]
    at = {(r["variant"], round(r["ebn0"], 2)): r for r in s}
    return ("| the reference's other five traces, `refs/TX_RX/*.txt` (its full chain, sample-serial synchronizers included): the filtered loop with genie timing, and with the in-scope "
            "synchronizers doing the work | filtered / baseband FER %.2f-%.2f over %d points; frame synchronizer in the loop: FER %.4f against the genie's %.4f at 3.8 dB; L&R + pilot-aided phase: %.4f "
            "(0.06 dB) | the reference's traces: 0.0197-0.0355 at 3.8 dB (0.07-0.09 dB from the genie curve) | `results/r06/filtered_loop.md`; `test_filtered_loop_matches_the_baseband_loop_and_stays_below_the_full_chain_traces`, "
            "`test_in_scope_synchronizers_in_the_loop_at_the_operating_point` |" % (min(rat), max(rat), len(rat), at[("frame", 3.8)]["fer"], [r for r in g if round(r["ebn0"], 2) == 3.8][0]["fer"], at[("fine", 3.8)]["fer"]))


def render(tag, bench_path=None):
    import refs_pooled
    p = bench_path or os.path.join(ROOT, "profiles", "%s_bench.json" % tag)
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    ro, ex = d["roofline"], d["extra"]
    c2, c3, c4 = ex["configs"]["2"], ex["configs"]["3"], ex["configs"]["4"]
    f1 = c4["per_F"][0]
    sp, rc, sl, cb = ex["spa"], ex["ref_config"], ex["sync_located"], d.get("cpu_baseline") or {}
    rd = os.path.join(ROOT, "results", tag)
    pq, pn, pe = refs_pooled.pooled(rd, "r06_clip_"), refs_pooled.pooled(rd, "r06_tanhnat_"), refs_pooled.pooled(rd, "r06_exact_")
    big = c4["per_F"][-1]
    L = [BEGIN, "",
         "One MI355X, `python bench.py` (`profiles/%s_bench.json`); parity rows from `results/%s/spa_rules.md`.  Regenerate: `python tools/make_readme_table.py %s --write`." % (tag, tag, tag), "",
         "| what | figure | against its bound | pinned by |", "|---|---|---|---|",
         "| **BASELINE metric** -- configs[1]: LDPC layered NMS, N = 64800 8/9, 10 ite, 4096 frames in HBM | **%.2f Gb/s = %.0f k frames/s** (%.3f ms per step) | counter bytes / time = %.2f of the 8 TB/s HBM peak (`roofline.frac`; SURVEY 8(d) algorithmic figure %.2f x); vector issue %s | `test_ldpc_baseline_batch_of_exactly_4096_frames_matches_oracle` (bit-exact) |" % (
             d["value"] / 1e9, d["fec_frames_per_s"] / 1e3, d["ms_per_step"], ro["frac"] or 0, ro["algorithmic_frac"], ("%.2f" % ro["resources"]["vector_issue"]["frac"]) if ro["resources"]["vector_issue"]["frac"] else "n/a"),
         "| configs[2]: fused RX chain QPSK-N 8/9 (PL frames -> info bits) | %.3f ms per 4096 = %.2f Gb/s | %.3f x its floor (LDPC launch + front-end bytes at the measured copy rate) | `tests/test_chain_gpu.py` |" % (c2["ms"], c2["info_bits_per_s"] / 1e9, c2["tail_over_floor"]),
         "| configs[3]: 16APSK-N 8/9, NMS 20 ite, fused chain | %.3f ms per 4096 = %.2f Gb/s | %.3f x its floor | `tests/test_chain_gpu.py` |" % (c3["ms"], c3["info_bits_per_s"] / 1e9, c3["tail_over_floor"]),
         "| configs[4]: 32APSK-S 3/4 behind the 81-tap matched filter, one frame per call | %.3f ms (early stop: %.3f ms); 4096 frames: %.2f M frames/s | one workgroup's 120 layers of 1.4 us; FIR %.0f fp32-equivalent TFLOP/s at 4096 frames | `tests/test_rx_lite_gpu.py`, `tests/test_fir_gpu.py` |" % (
             f1["latency_ms_median"], f1.get("latency_ms_early_stop") or 0, big["frames_per_s"] / 1e6, (big["fir_GFLOPs_fp32_equiv"] or 0) / 1e3),
         "| the reference's default decoder, `--dec-implem SPA`, 10 ite fixed | N = 64800: %.0f k frames/s; N = 16200: %.2f M frames/s | N = 16200: vector issue 0.97 busy (a quarter of it transcendentals) with the fp32 messages streaming at 0.76 of the fabric's rate; N = 64800: 0.78 / 0.73 (`profiles/r06_ldpc_variants.md`) | `test_ldpc_spa_matches_oracle` (1e-4 max(1, abs(L))) |" % (
             sp["QPSK-N_8/9"]["fec_frames_per_s"] / 1e3, sp["QPSK-S_8/9"]["fec_frames_per_s"] / 1e6),
         "| the reference's own configuration: QPSK-S 8/9, SPA 50 ite, early stop, 3.8 dB, TX + AWGN + RX + monitor on the GPU, -F 8192 | %.1f Gb/s (1 clone), **%.1f Gb/s** (3 clones) | the reference's trace: 24.5 Mb/s on an unstated CPU | `tests/test_refs_gpu.py` (19 rows + pooled) |" % (
             rc["clones_1"]["info_Gbps"], rc["clones_3"]["info_Gbps"]),
         "| **parity with the reference** (pooled FER over its 19 regression rows, run / reference) | `SPA` %.3f +- %.3f; the reference's decoder as recalled (`SPA_TANH`, natural order) **%.3f +- %.3f**; rounds 1-5's rule %.3f +- %.3f | chi^2 on 19 dof: %.1f / %.1f / %.1f | `test_gpu_spa50_pooled_over_the_19_rows_and_the_three_rules` |" % (
             pq["pooled_ratio"], pq["pooled_ratio"] * pq["pooled_sigma"], pn["pooled_ratio"], pn["pooled_ratio"] * pn["pooled_sigma"], pe["pooled_ratio"], pe["pooled_ratio"] * pe["pooled_sigma"],
             pq["chi2"], pn["chi2"], pe["chi2"]) if pq and pn and pe else "| parity with the reference | (results missing) | | |",
         floor_row(rd),
         loop_row(rd),
         "| natural row order (the reference's sweep), NMS 10 ite, 4096 normal frames | %.0f k frames/s | bit-exact with the oracle's ORC_SCHED_NATURAL | `test_ldpc_natural_order_matches_oracle` |" % ((ex.get("natural_order_fps") or 0) / 1e3),
         "| frame synchronizer, located form | %.3f ms per 4096 32APSK-S frames, %.3f ms per 1024 QPSK-N frames | %.2f / %.2f of 8 TB/s (16 B per sample) | `tests/test_sync_gpu.py` |" % (
             sl["32APSK-S_3/4"]["ms_per_call"], sl["QPSK-N_8/9"]["ms_per_call"], sl["32APSK-S_3/4"]["frac_of_8TBps"], sl["QPSK-N_8/9"]["frac_of_8TBps"]),
         "| CPU baseline in the same run (oracle port, inter-frame SIMD, pinned threads) | %s | GPU / CPU = %s | `bench.py` `cpu_baseline` |" % (
             ("%.1f k frames/s on %d threads" % (cb["fec_frames_per_s"] / 1e3, cb["cores"])) if cb.get("fec_frames_per_s") else "not in this run", ("%.0f x" % (d["fec_frames_per_s"] / cb["fec_frames_per_s"])) if cb.get("fec_frames_per_s") else "-"),
         "", END]
    return "\n".join(L)


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r06"
    bp = sys.argv[sys.argv.index("--bench") + 1] if "--bench" in sys.argv else None
    blk = render(tag, bp)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "README.md")
        s = open(p).read()
        a, b = s.index(BEGIN), s.index(END) + len(END)
        open(p, "w").write(s[:a] + blk + s[b:])
    else:
        print(blk)
