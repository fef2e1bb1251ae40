"""README.md's headline table and results/r06/spa_rules.md quote no hand-typed figure: both are the output of their generators over committed measurement files
(profiles/r06_bench.json, results/r06/r06_*.txt, floor_spa.txt).  And tools/refs_pooled.py's arithmetic on a case that can be checked by hand."""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_readme_table_is_the_generators_output():
    import make_readme_table as M
    s = open(os.path.join(ROOT, "README.md")).read()
    a, b = s.index(M.BEGIN), s.index(M.END) + len(M.END)
    assert s[a:b] == M.render("r06"), "README.md's generated table is stale or hand-edited: python tools/make_readme_table.py r06 --write"
    assert "pooled FER" in s[a:b] and "no error floor" in s[a:b] and s[:a].count("\n") < 12          # the table is what a reader meets first


def test_spa_rules_md_is_the_generators_output():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_spa_rules_md.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "results", "r06", "spa_rules.md")).read(), "python tools/make_spa_rules_md.py > results/r06/spa_rules.md"
    # what DESIGN.md section 2 quotes from it
    import refs_pooled
    d = os.path.join(ROOT, "results", "r06")
    exact, clip, tanhnat = refs_pooled.pooled(d, "r06_exact_"), refs_pooled.pooled(d, "r06_clip_"), refs_pooled.pooled(d, "r06_tanhnat_")
    assert exact["n"] == clip["n"] == tanhnat["n"] == 19
    assert abs(exact["pooled_ratio"] - 1.080) < 5e-4 and exact["z"] > 3.5 and exact["chi2"] > 45
    assert abs(clip["pooled_ratio"] - 1.047) < 5e-4 and clip["chi2"] < 20
    assert abs(tanhnat["pooled_ratio"] - 1.020) < 5e-4 and abs(tanhnat["z"]) < 1.5 and abs(tanhnat["common_slope"]) < tanhnat["common_slope_sigma"]


def test_refs_pooled_arithmetic(tmp_path):
    """two rows of one trace with known counts: the pooled log-ratio is the 1 / sigma^2-weighted mean of the rows' log-ratios, sigma^2 = 1 / FE_ref + 1 / FE_run"""
    import json
    import refs_pooled
    refs = json.load(open(os.path.join(ROOT, "tests", "golden", "refs_tx_rx_bb.json")))
    rows = refs["QPSK_8_9.txt"]["rows"][:2]
    lines = []
    want = []
    for r, mult, fe in zip(rows, (1.10, 0.95), (4000, 2500)):
        fra = int(round(fe / (mult * r["fe"] / r["fra"])))
        lines.append("  %9.2f | %8.2f || %8d | %8d | %8d | %8.2e | %8.2e || %8.3f | 00h00'00" % (r["esn0"], r["ebn0"], fra, 10 * fe, fe, 0.0, fe / fra, 1.0))
        lr = math.log((fe / fra) / (r["fe"] / r["fra"]))
        want.append((lr, 1.0 / r["fe"] + 1.0 / fe))
    open(tmp_path / "t_qpsk_8_9.txt", "w").write("\n".join(lines) + "\n")
    res = refs_pooled.pooled(str(tmp_path), "t_")
    w = [1 / v for _, v in want]
    m = sum(l * wi for (l, _), wi in zip(want, w)) / sum(w)
    assert res["n"] == 2 and abs(res["pooled_log_ratio"] - m) < 1e-12 and abs(res["pooled_sigma"] - 1 / math.sqrt(sum(w))) < 1e-12
    assert abs(res["chi2"] - sum(l * l / v for l, v in want)) < 1e-9
