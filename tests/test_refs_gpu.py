"""Statistical pin on the reference's own regression traces (refs/TX_RX_BB/*.txt -- the only result-pinning
artefacts the reference ships: SPA, 50 iterations) with the whole Monte-Carlo loop on the GPU: FER inside
the CI's x2.5 sensibility band (.gitlab-ci.yml:117).  One row per reference MODCOD here; all 19 rows are
replayed by tools/compare_refs.py (results/r01/refs_comparison.md)."""
import io
import json
import os

import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROWS = [("QPSK_8_9.txt", 3.7), ("QPSK_3_5.txt", 1.4), ("8PSK_3_5.txt", 2.9), ("8PSK_8_9.txt", 6.4), ("16APSK_8_9.txt", 7.3)]


@pytest.mark.parametrize("ref,ebn0", ROWS)
def test_gpu_spa50_fer_inside_reference_band(ref, ebn0):
    from dvbs2_amd import sim
    refs = json.load(open(os.path.join(GOLD, "refs_tx_rx_bb.json")))
    d = refs[ref]
    row = [r for r in d["rows"] if abs(r["ebn0"] - ebn0) < 1e-6][0]
    argv = ["--mod-cod", d["header"]["modcod"], "-m", "%.2f" % ebn0, "-M", "%.2f" % (ebn0 + 0.01), "--dec-implem", "SPA",
            "--dec-ite", "50", "-F", "2048", "--max-frames", "200000", "-e", "100"]
    if "PERFECT" in d["command"]:
        argv += ["--est-type", "PERFECT"]
    r = sim.run(sim.build_parser().parse_args(argv), out=io.StringIO())[0]
    assert r["fe"] >= 100
    assert row["fer"] / 2.5 <= r["fer"] <= row["fer"] * 2.5, (r["fer"], row["fer"])
    assert row["ber"] / 2.5 <= r["ber"] <= row["ber"] * 2.5, (r["ber"], row["ber"])
    assert abs(r["esn0"] - row["esn0"]) < 0.0051
