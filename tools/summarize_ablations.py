"""tools/run_ablations.sh's output -> profiles/ldpc_ablation.json: what the headline kernel gives back when a part of its layer is left out (timing-only builds, -DW8_ABL),
stamped with the hash of the kernel + plan sources it was measured on.  bench.py's `roofline` reads it: the launch with BOTH resources' work removed (W8_ABL=15: no global slot
traffic, a third of the layer's vector instructions gone) is the dependent chain's own time -- `chain_floor_ms` -- and the elasticities say whether a resource binds.
usage: python tools/summarize_ablations.py gpurun_out/r05_ablations.txt [results/r05/ablations.txt]"""
import json, os, re, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

WHAT = {1: "pass 2's 12 global stores per check dropped", 2: "pass 1a's 12 global loads per check replaced by a register move", 3: "no global slot traffic in the layer (1 + 2)",
        4: "pass 1b without the min1 / min2 / sign tracking (-20 % of the layer's vector instructions)", 8: "pass 2 without the compare and the selects (-15 %)",
        12: "both arithmetic cuts (-35 % of the layer's vector instructions)", 15: "everything above: no global slot traffic and -35 % of the vector instructions"}
VALU_CUT = {1: 0.0, 2: 0.0, 3: 0.0, 4: 0.20, 8: 0.15, 12: 0.35, 15: 0.35}

def main():
    src = sys.argv[1]
    ms = {}
    for l in open(src):
        m = re.match(r"(libdvbs2hip|lib_abl(\d+))\.so N:\s+(\d+) frames\s+([\d.]+) ms", l)
        if m:
            ms.setdefault(int(m.group(2)) if m.group(2) else 0, []).append(float(m.group(4)))
    if 0 not in ms or 15 not in ms:
        raise SystemExit("no production / W8_ABL=15 rows in " + src)
    mean = {k: statistics.mean(v) for k, v in ms.items()}
    prod = mean[0]
    out = {"kernel_sha": bench.kernel_sha(), "frames": 4096, "n_ite": 10, "modcod": "QPSK-N_8/9", "source": os.path.basename(src), "rounds": len(ms[0]),
           "what": "same-box alternation of timing-only builds of k_ldpc_wg8.hip (-DW8_ABL=n, wrong results by construction) with the production library; wall ms per launch of the bits socket (tools/scan_batch.py, mean of 7), mean over the rounds",
           "production_ms": prod, "chain_floor_ms": mean[15],
           "ablations": {str(k): {"what": WHAT[k], "ms": mean[k], "change": mean[k] / prod - 1.0, "vector_instructions_cut": VALU_CUT[k]} for k in sorted(mean) if k},
           # elasticity = relative time given back per relative amount of the resource's work removed
           "elasticity": {"vector_issue": (1.0 - mean[12] / prod) / 0.35 if 12 in mean else None,
                          "global_slot_traffic": (1.0 - mean[3] / prod) / 1.0 if 3 in mean else None}}
    dst = os.path.join(ROOT, "profiles", "ldpc_ablation.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        os.makedirs(os.path.dirname(sys.argv[2]), exist_ok=True)
        open(sys.argv[2], "w").write(open(src).read())

if __name__ == "__main__":
    main()
