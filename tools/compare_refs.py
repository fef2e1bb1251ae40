"""Replays the reference's regression traces (refs/TX_RX_BB/*.txt, SPA 50 ite; rows committed in
tests/golden/refs_tx_rx_bb.json) on the GPU and compares FER/BER row by row with the CI's
sensibility band (x2.5, .gitlab-ci.yml:117).  GPU box only.  usage: python tools/compare_refs.py [out.md]"""
import io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvbs2_amd import sim

refs = json.load(open(os.path.join(ROOT, "tests", "golden", "refs_tx_rx_bb.json")))
lines = ["# GPU (SPA, 50 ite, QC-layer schedule) vs refs/TX_RX_BB (AFF3CT SPA, 50 ite, natural order)", "",
         "| ref file | MODCOD | Eb/N0 | ref FER | GPU FER | ratio | ref BER | GPU BER | frames | in x2.5 band |", "|---|---|---|---|---|---|---|---|---|---|"]
ok_all = True
for name, d in refs.items():
    if name.endswith("_inter.txt"):
        continue                     # same numbers as QPSK_8_9.txt (run with -F 2)
    mc = d["header"]["modcod"]
    for row in d["rows"]:
        argv = ["--mod-cod", mc, "-m", "%.2f" % row["ebn0"], "-M", "%.2f" % (row["ebn0"] + 0.01), "--dec-implem", "SPA", "--dec-ite", "50",
                "-F", "2048", "--max-frames", "400000", "-e", "100"]
        if "PERFECT" in d["command"]:
            argv += ["--est-type", "PERFECT"]
        args = sim.build_parser().parse_args(argv)
        r = sim.run(args, out=io.StringIO())[0]
        ratio = r["fer"] / row["fer"] if row["fer"] > 0 else float("nan")
        ok = 1 / 2.5 <= ratio <= 2.5
        ok_all &= ok
        lines.append("| %s | %s | %.2f | %.2e | %.2e | %.2f | %.2e | %.2e | %d | %s |" % (name, mc, row["ebn0"], row["fer"], r["fer"], ratio,
                                                                                  row["ber"], r["ber"], r["fra"], "yes" if ok else "NO"))
        print(lines[-1], flush=True)
lines += ["", "all rows inside the band: %s" % ok_all]
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "refs_comparison.md")
open(out, "w").write("\n".join(lines) + "\n")
