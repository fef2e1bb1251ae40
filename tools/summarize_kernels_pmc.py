#!/usr/bin/env python3
"""Condenses tools/profile_kernels.sh's rocprofv3 outputs (gpurun_out/pk_*) into profiles/<tag>_kernels_pmc.md: per kernel of the
path (other than the LDPC decoder) and per workload size -- launches of one kernel name are split by grid size -- the average
duration, calibrated fabric bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes; MI355X_MICROARCH.md HBM section), instruction
counts and the busy / wait shares that say what bounds it."""
import collections, csv, glob, hashlib, json, os, sys
sys_path_fix = __import__('sys').path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import kernel_mix as KM
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def first(pattern):
    g = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return g[-1] if g else None


def short(n):
    n = n.replace("void ", "").replace("dvbs2::", "")
    return n.split("(")[0]


want = ("front", "bch_decode", "sync_", "sff_", "vd_dmax", "demod", "monitor", "fir_")
dur = collections.defaultdict(list)
kt = first("pk_stats/*/*_kernel_trace.csv")
if kt:
    for r in csv.DictReader(open(kt)):
        k = short(r["Kernel_Name"])
        if any(w in k for w in want):
            # the counter files carry the TOTAL grid size, the kernel trace its three dimensions: the key is the product (a 2-D launch otherwise finds no counter row:
            # the all-zero sync_vdelay_batch_kernel rows of rounds 4-5)
            grid = r["Grid_Size"] if "Grid_Size" in r else str(int(r.get("Grid_Size_X", 1)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)))
            dur[(k, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pk_fetch", "pk_write", "pk_sq1", "pk_sq2", "pk_sq3"):
    f = first(d + "/*/*_counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(w in k for w in want):
            pmc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
STAMPED = ("k_front.hip", "k_bch.hip", "k_sync.hip", "k_sync_mfma.hip", "k_fir.hip", "k_fir_mfma.hip", "k_tx.hip")


def sources_sha():
    h = hashlib.sha256()
    for f in STAMPED:
        h.update(open(os.path.join(ROOT, "dvbs2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


lines = ["# rocprofv3 counters of the non-LDPC kernels (%s) -- `python3 tools/pmc_workload.py`, passes of tools/profile_kernels.sh" % tag, "",
         "Sources measured: %s, sha-256 `%s` (profiles/kernels_pmc_stamp.json; tests/test_bench_contract.py fails when they change without a new profile)." % (", ".join(STAMPED), sources_sha()), "",
         "Per kernel and grid size (= workload): average over its launches.  fabric GB = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE",
         "reports half of the bytes of a coalesced stream on gfx950, WRITE_SIZE the bytes: calibrated in profiles/r02_ldpc_rocprof.md).",
         "VALU busy = SQ_ACTIVE_INST_VALU / (4 SIMDs x SQ_BUSY_CYCLES summed over the chip's SQs) is shown as the share of wave-cycles instead:",
         "valu/wave-cyc = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES, trans = v_exp / v_log / v_rcp / v_sqrt instructions.", "",
         "VALU issue = (non-transcendental SQ_INSTS_VALU x the price of the kernel's static instruction mix + trans x 8.06) / (1024 SIMDs x SQ_BUSY_CYCLES / 32).  Prices (round 4, profiles/r04_probe_issue.txt, tools/kernel_mix.py): 2.07 SIMD cycles per instruction of the simple two-operand class (v_mov / v_and / v_or / v_xor / v_add / v_sub / v_mul with register or inline operands), 2.6 the same with a literal, 4.2-4.25 for everything else (VOP3 / VOP3P / SDWA / DPP encodings, SGPR or vcc operands, VOPC, v_min / v_max / shifts / conversions / v_fmac), v_exp / v_log / v_rcp 8.06 -- whatever the number of waves (round 3 priced everything at 2, round 2 at 4; the first pricing of round 4 had VOPC, v_cndmask_b32_e32, v_min / v_max and the shifts in the 2.07 class: probe_issue4 put them at 4.25).", "",
         "| kernel | grid | launches | avg us | fabric GB | fabric TB/s | VALU inst | trans inst | VALU issue | LDS inst | VMEM rd / wr | valu / wave-cyc | wait / wave-cyc | LDS bank-conflict / LDS active | L2 hit |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
rows_json = []
for key in sorted(dur, key=lambda k: -sum(dur[k])):
    c = {n: sum(v) / len(v) for n, v in pmc.get(key, {}).items()}
    us = sum(dur[key]) / len(dur[key])
    fab = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
    wc = c.get("SQ_WAVE_CYCLES", 0)
    def ratio(a, b): return "%.2f" % (a / b) if b else "-"
    hit = c.get("TCC_HIT_sum", 0); miss = c.get("TCC_MISS_sum", 0)
    busy = c.get("SQ_BUSY_CYCLES", 0) / 32.0
    cpi, how = KM.price_kernel(key[0])
    tr = c.get("SQ_INSTS_VALU_TRANS_F32", 0)
    issue = ((c.get("SQ_INSTS_VALU", 0) - tr) * (cpi or 3.0) + tr * 8.06) / (1024.0 * busy) if busy else None
    if not c:      # the counter passes attributed no row to this (kernel, grid) -- e.g. a launch whose grid differs between the passes: say so instead of printing zeros
        rows_json.append(dict(kernel=key[0], grid=key[1], launches=len(dur[key]), avg_us=us, fabric_bytes=None, valu_issue=None, cycles_per_valu=cpi, valu_mix=how, counters=None))
        lines.append("| `%s` | %s | %d | %.1f | n/a: the counter passes attributed nothing to this launch | | | | | | | | | | |" % (key[0], key[1], len(dur[key]), us))
        continue
    rows_json.append(dict(kernel=key[0], grid=key[1], launches=len(dur[key]), avg_us=us, fabric_bytes=fab, valu_issue=issue, cycles_per_valu=cpi, valu_mix=how, counters=c))
    lines.append("| `%s` | %s | %d | %.1f | %.3f | %.2f | %.3g | %.3g | %s | %.3g | %.3g / %.3g | %s | %s | %s | %s |" % (
        key[0], key[1], len(dur[key]), us, fab / 1e9, fab / (us * 1e-6) / 1e12 if us else 0, c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_VALU_TRANS_F32", 0),
        ("%.2f" % issue) if issue is not None else "-", c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_VMEM_WR", 0), ratio(c.get("SQ_ACTIVE_INST_VALU", 0), wc), ratio(c.get("SQ_WAIT_ANY", 0), wc),
        ratio(c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_LDS_IDX_ACTIVE", 0)), ratio(hit, hit + miss)))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "%s_kernels_pmc.md" % tag), "w").write("\n".join(lines) + "\n")
json.dump(dict(tag=tag, sources=list(STAMPED), sha=sources_sha(), rows=rows_json), open(os.path.join(ROOT, "profiles", "%s_kernels_pmc.json" % tag), "w"), indent=1)
json.dump(dict(tag=tag, sources=list(STAMPED), sha=sources_sha(), file="profiles/%s_kernels_pmc.md" % tag), open(os.path.join(ROOT, "profiles", "kernels_pmc_stamp.json"), "w"), indent=1)
print("\n".join(lines))
