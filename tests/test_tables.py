"""Host logic: MODCOD table, frame sizes, LDPC address-table invariants."""
import collections
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_modcods_sizes(P):
    # Appendix A of SURVEY.md / DVBS2.cpp:287-356
    want = {"QPSK-S_8/9": (14232, 14400, 16200, 2, 8100, 8370), "QPSK-S_3/5": (9552, 9720, 16200, 2, 8100, 8370),
            "8PSK-S_3/5": (9552, 9720, 16200, 3, 5400, 5598), "8PSK-S_8/9": (14232, 14400, 16200, 3, 5400, 5598),
            "16APSK-S_8/9": (14232, 14400, 16200, 4, 4050, 4212)}
    for name, (kb, nb, nl, bps, nx, plf) in want.items():
        mc = P.get_modcod(name)
        assert (mc.K_bch, mc.N_bch, mc.N_ldpc, mc.bps, mc.N_xfec, mc.pl_frame) == (kb, nb, nl, bps, nx, plf)
        assert mc.in_reference
    assert P.get_modcod("").name == "QPSK-S_8/9"                       # DVBS2.cpp:290
    with pytest.raises(ValueError, match="mod-cod scheme not supported"):  # DVBS2.cpp:319
        P.get_modcod("QPSK-S_1/2")
    # extension rows
    assert P.get_modcod("QPSK-N_8/9").pl_frame == 33282 and P.get_modcod("16APSK-N_8/9").pl_frame == 16686
    assert P.get_modcod("32APSK-S_3/4").pl_frame == 3402 and P.get_modcod("32APSK-S_3/4").K_bch == 11712


def test_sizes_match_refs_headers(P):
    refs = json.load(open(os.path.join(GOLD, "refs_tx_rx_bb.json")))
    # "N. cw (N)" of the radio header = 2 * pl_frame * osf(2): 33480 QPSK, 16848 16APSK
    for fname, d in refs.items():
        mc = P.get_modcod(d["header"]["modcod"])
        assert int(d["header"]["n_cw"]) == 2 * mc.pl_frame * 2
        for row in d["rows"]:       # Es/N0 column = ebn0_to_esn0 (TX_RX_BB/main.cpp:142-146)
            assert abs(P.ebn0_to_esn0(row["ebn0"], mc.code_rate, mc.bps) - row["esn0"]) < 0.0051


def test_full_chain_traces_are_the_same_noise_points_as_the_baseband_ones(P):
    """refs/TX_RX/*.txt (the reference's full chain: its own synchronizers, not a genie): same MODCOD, same Es/N0 <-> Eb/N0 mapping (TX_RX/main.cpp:408) and the same stopping rule as
    the baseband traces, so the two sets can be read side by side; at every common point the full chain loses at least as many frames as the baseband loop."""
    full = json.load(open(os.path.join(GOLD, "refs_tx_rx.json")))
    bb = {round(r["ebn0"], 2): r for r in json.load(open(os.path.join(GOLD, "refs_tx_rx_bb.json")))["QPSK_8_9.txt"]["rows"]}
    assert len(full) == 5
    for fname, d in full.items():
        mc = P.get_modcod(d["header"]["modcod"])
        assert mc.name == "QPSK-S_8/9" and d["header"]["perfect_sync"] == "NO" and d["header"]["implem"] == "SPA" and d["header"]["n_ite"] == "50"
        for row in d["rows"]:
            assert abs(P.ebn0_to_esn0(row["ebn0"], mc.code_rate, mc.bps) - row["esn0"]) < 0.0051 and 100 <= row["fe"] <= 101
            b = bb.get(round(row["ebn0"], 2))
            if b:
                assert row["fer"] > 1.5 * b["fer"], (fname, row["ebn0"])


@pytest.mark.parametrize("name,N,K,E,deg", [("N16200_8_9.txt", 16200, 14400, 48599, {4: 5, 3: 35}),
                                            ("N16200_3_5.txt", 16200, 9720, 71279, {12: 9, 3: 18}),
                                            ("N16200_3_4.txt", 16200, 11880, 47519, {12: 1, 3: 32}),
                                            ("N64800_8_9.txt", 64800, 57600, 194399, {4: 20, 3: 140})])
def test_ldpc_table_invariants(P, name, N, K, E, deg):
    rp, ad = P.load_ldpc_table(name)
    M = N - K
    q = M // 360
    assert len(rp) - 1 == K // 360 and ad.min() >= 0 and ad.max() < M
    assert dict(collections.Counter(np.diff(rp).tolist())) == deg
    assert len(ad) * 360 + 2 * M - 1 == E                                # SURVEY.md 8(d) edge counts
    rows = [ad[rp[i]:rp[i + 1]] for i in range(len(rp) - 1)]
    assert all(len(set(r.tolist())) == len(r) for r in rows)             # no duplicate edge
    if name != "N16200_3_4.txt":                                         # regular codes: uniform check degree
        res = collections.Counter((ad % q).tolist())
        assert set(res.values()) == {len(ad) // q}
    # no length-4 cycles between bit-groups: two groups never share two (layer, relative shift) pairs
    seen = {}
    for g, r in enumerate(rows):
        for a in r.tolist():
            seen.setdefault(a % q, []).append((g, a // q))
    for layer, lst in seen.items():
        diffs = collections.Counter()
        for i in range(len(lst)):
            for j in range(i + 1, len(lst)):
                (g1, s1), (g2, s2) = lst[i], lst[j]
                if g1 != g2:
                    diffs[(g1, g2, (s1 - s2) % 360)] += 1
        # the same (g1, g2, shift difference) in two different layers would close a 4-cycle
        seen[layer] = set(diffs)
    allp = collections.Counter(k for s in seen.values() for k in s)
    assert max(allp.values()) == 1


def test_rrc_taps_properties(P):
    t = P.rrc_taps(0.2, 2, 20)
    assert t.size == 81 and abs(float((t.astype(np.float64) ** 2).sum()) - 1.0) < 1e-6
    assert np.array_equal(t, t[::-1])
    full = np.convolve(t.astype(np.float64), t.astype(np.float64))      # RRC * RRC = Nyquist
    c = full.size // 2
    assert abs(full[c] - 1.0) < 1e-6 and np.max(np.abs(full[c + 2::2][:30])) < 2e-3


def test_shard_helpers():
    from dvbs2_amd.parallel import shard_frames, shard_stream
    for n, w in ((4096, 8), (10, 3), (7, 8)):
        parts = [shard_frames(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
    lo, hi, hlo = shard_stream(1000, 1, 4, 80)
    assert (lo, hi, hlo) == (250, 500, 170)
