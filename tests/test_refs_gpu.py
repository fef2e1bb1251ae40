"""Statistical pin on the reference's own regression traces (refs/TX_RX_BB/*.txt -- the only result-pinning
artefacts the reference ships: SPA, 50 iterations) with the whole Monte-Carlo loop on the GPU: FER inside
the CI's x2.5 sensibility band (.gitlab-ci.yml:117).  ALL 19 rows of the five distinct traces, each to >= 100 frame
errors like the reference's own -e 100 (a table of one such run: results/r01/refs_comparison.md; the CPU oracle's own
pin on the same rows: results/r02/oracle_refs_pin.md).  Plus the one external anchor the N = 64800 extension can have:
ETSI EN 302 307 Table 13."""
import io
import json
import os

import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_REFS = json.load(open(os.path.join(GOLD, "refs_tx_rx_bb.json")))
ROWS = [(name, r["ebn0"]) for name, d in _REFS.items() if not name.endswith("_inter.txt") for r in d["rows"]]      # _inter = the same rows with -F 2
assert len(ROWS) == 19


@pytest.mark.parametrize("ref,ebn0", ROWS)
def test_gpu_spa50_fer_inside_reference_band(ref, ebn0):
    from dvbs2_amd import sim
    refs = json.load(open(os.path.join(GOLD, "refs_tx_rx_bb.json")))
    d = refs[ref]
    row = [r for r in d["rows"] if abs(r["ebn0"] - ebn0) < 1e-6][0]
    argv = ["--mod-cod", d["header"]["modcod"], "-m", "%.2f" % ebn0, "-M", "%.2f" % (ebn0 + 0.01), "--dec-implem", "SPA",
            "--dec-ite", "50", "-F", "2048", "--max-frames", "400000", "-e", "100"]
    if "PERFECT" in d["command"]:
        argv += ["--est-type", "PERFECT"]
    r = sim.run(sim.build_parser().parse_args(argv), out=io.StringIO())[0]
    assert r["fe"] >= 100
    assert row["fer"] / 2.5 <= r["fer"] <= row["fer"] * 2.5, (r["fer"], row["fer"])
    assert row["ber"] / 2.5 <= r["ber"] <= row["ber"] * 2.5, (r["ber"], row["ber"])
    assert abs(r["esn0"] - row["esn0"]) < 0.0051


@pytest.mark.parametrize("clones", [2, 3])
def test_gpu_sim_with_clones_counts_what_one_clone_counts(clones):
    """--clones C (the reference's Sequence with n_threads clones, TX_RX_BB/main.cpp:19,96): the batches are the same batches -- dealt to C handles in turn -- so the counters after
    the same number of batches are the one-clone loop's: a run capped at 6 batches (no row reaches 100 frame errors before) gives FRA / BE / FE equal to the one-clone run's."""
    from dvbs2_amd import sim
    def run(c):
        argv = ["--mod-cod", "QPSK-S_8/9", "-m", "3.90", "-M", "3.91", "--dec-implem", "NMS", "--dec-ite", "10", "-F", "1024", "--max-frames", str(6 * 1024), "-e", "1000000",
                "--clones", str(c)]
        return sim.run(sim.build_parser().parse_args(argv), out=io.StringIO())[0]
    one, many = run(1), run(clones)
    # with C clones the loop stops when the collected total reaches the cap: batches issued = 6 + (C - 1) in flight at that moment, all counted
    assert one["fra"] == 6 * 1024 and many["fra"] == (6 + clones - 1) * 1024
    assert many["fe"] >= one["fe"] > 0 and many["be"] >= one["be"]
    capped = ["--mod-cod", "QPSK-S_8/9", "-m", "3.90", "-M", "3.91", "--dec-implem", "NMS", "--dec-ite", "10", "-F", "1024", "--max-frames", str((6 + clones - 1) * 1024), "-e", "1000000"]
    same = sim.run(sim.build_parser().parse_args(capped), out=io.StringIO())[0]
    assert (same["fra"], same["be"], same["fe"]) == (many["fra"], many["be"], many["fe"])      # the same batches, seed for seed


@pytest.mark.parametrize("modcod,anchor_db", [("QPSK-N_8/9", 6.20), ("8PSK-N_8/9", 10.69), ("16APSK-N_8/9", 12.89)])
def test_normal_frame_waterfall_sits_at_the_etsi_anchor(modcod, anchor_db):
    """The N = 64800 codes are an extension beyond the reference (their LDPC table is entered from ETSI EN 302 307 Annex B): the only
    external numbers they can be held against are the standard's own -- EN 302 307-1 Table 13, rate 8/9 normal FECFRAME, Es/N0 for
    quasi-error-free operation (PER 1e-7, 50 iterations, ideal demodulation): QPSK 6.20 dB, 8PSK 10.69 dB, 16APSK 12.89 dB.  With SPA, 50
    iterations, the waterfall must sit right there: at the anchor frames essentially never fail (FER < 2e-3 over 40 960 frames; measured
    1e-5 .. 2e-5 for all three, results/r02/etsi_anchors.txt), 0.3 dB below it most do.  A wrong address in the table, a wrong bit order in a
    constellation or a wrong interleaver moves the curve by far more."""
    from dvbs2_amd import sim
    from dvbs2_amd import params as P
    import math
    mc = P.get_modcod(modcod)
    def run(esn0, max_frames):
        ebn0 = esn0 - 10.0 * math.log10(mc.bps * mc.K_bch / mc.N_ldpc)
        argv = ["--mod-cod", modcod, "-m", "%.3f" % ebn0, "-M", "%.3f" % (ebn0 + 0.001), "--dec-implem", "SPA", "--dec-ite", "50", "-F", "4096",
                "--max-frames", str(max_frames), "-e", "100"]
        if mc.bps >= 4:
            argv += ["--est-type", "PERFECT"]          # like the reference's own APSK trace: M2M4 assumes a constant modulus
        r = sim.run(sim.build_parser().parse_args(argv), out=io.StringIO())[0]
        assert abs(r["esn0"] - esn0) < 0.006
        return r
    at = run(anchor_db, 40960)
    assert at["fra"] >= 40960 and at["fer"] < 2e-3, at
    below = run(anchor_db - 0.30, 8192)
    assert below["fer"] > 0.5, below
