#!/usr/bin/env python3
"""gpurun_out/ref_config_*.txt (tools/ref_config_spa50.sh: the reference's own command lines through the C++ simulator on the GPU) -> the table rows of
results/<tag>/ref_config_spa50.md: GPU FER / BER / SIM_THR beside the reference's trace.  usage: python tools/summarize_ref_config.py r03"""
import json, os, re, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
refs = json.load(open(os.path.join(ROOT, "tests", "golden", "refs_tx_rx_bb.json")))
files = [("QPSK_8_9.txt", "QPSK-S_8/9", "qpsk_8_9"), ("QPSK_3_5.txt", "QPSK-S_3/5", "qpsk_3_5"), ("8PSK_3_5.txt", "8PSK-S_3/5", "8psk_3_5"),
         ("8PSK_8_9.txt", "8PSK-S_8/9", "8psk_8_9"), ("16APSK_8_9.txt", "16APSK-S_8/9", "16apsk_8_9")]
rows = []
for ref_file, modcod, name in files:
    gpu = {}
    for l in open(os.path.join(ROOT, "gpurun_out", os.environ.get("REF_CONFIG_PREFIX", "ref_config_") + "%s.txt" % name)):      # REF_CONFIG_PREFIX=ref1000_: the -e 1000 runs of tools/ref_config_e1000.sh
        if re.match(r"^ +[0-9]", l):
            f = [x.strip() for x in l.replace("||", "|").split("|")]
            gpu[round(float(f[1]), 2)] = dict(fra=int(f[2]), fe=int(f[4]), ber=float(f[5]), fer=float(f[6]), thr=float(f[7]))
    for r in refs[ref_file]["rows"]:
        eb = round(float(r["ebn0"]), 2)
        if eb in gpu:
            g = gpu[eb]
            rows.append("| %s | %s | %.2f | %.2e | %.2e | %.2e | %.2e | %.3f | %.0f | %d | %d |" % (ref_file, modcod, eb, r["fer"], g["fer"], r["ber"], g["ber"], r["thr_mbps"], g["thr"], g["fra"], g["fe"]))
print("\n".join(rows))
