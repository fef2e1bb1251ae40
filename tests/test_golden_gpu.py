"""GPU vs the COMMITTED golden fixtures (tests/golden, made by make_golden.py): the parity
claim does not depend on rebuilding the oracle on the GPU box."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


def test_ldpc_golden(Rx):
    k = np.load(os.path.join(GOLD, "kat_ldpc_short_8_9.npz"))
    rx = Rx("QPSK-S_8/9", max_frames=2, n_ite=int(k["n_ite"]), alpha=float(k["alpha"]), early_stop=False)
    V, CWD, post, _ = rx.decode_siho(k["llr"], with_post=True)
    assert np.array_equal(np.packbits(V.astype(np.uint8), axis=1), k["bits_qc"])
    assert np.array_equal(CWD, k["cwd_qc"])
    assert np.max(np.abs(post - k["post_qc"])) <= 1e-4 and np.array_equal(post, k["post_qc"])
    # the reference's natural row order (dvbs2hip_set_ldpc_schedule): the fixture holds the oracle's NATURAL results too
    from dvbs2_amd import lib_binding as B
    rx.set_ldpc_schedule(B.SCHED_NATURAL)
    V, CWD, post, _ = rx.decode_siho(k["llr"], with_post=True)
    assert np.array_equal(np.packbits(V.astype(np.uint8), axis=1), k["bits_nat"]) and np.array_equal(CWD, k["cwd_nat"])
    assert np.array_equal(post, k["post_nat"])
    rx.close()


def test_ldpc_normal_golden(Rx):
    """The headline code at size from COMMITTED data (N = 64800 rate 8/9, NMS, 10 iterations, alpha 1.0; one frame at 3.0 dB that does not
    converge, one at 4.2 dB that does): hard bits, CWD, iteration counts and the posteriors' bit patterns, QC and NATURAL order, fixed
    iterations and the stopping rule."""
    from dvbs2_amd import lib_binding as B
    k = np.load(os.path.join(GOLD, "kat_ldpc_normal_8_9.npz"))
    rx = Rx("QPSK-N_8/9", max_frames=2, n_ite=int(k["n_ite"]), alpha=float(k["alpha"]), early_stop=False)
    for tag, sched in (("qc", B.SCHED_QC), ("nat", B.SCHED_NATURAL)):
        rx.set_ldpc_schedule(sched)
        rx.set_ldpc_params(int(k["n_ite"]), float(k["alpha"]), False)
        V, CWD, post, ites = rx.decode_siho(k["llr"], with_post=True)
        assert np.array_equal(np.packbits(V.astype(np.uint8), axis=1), k["bits_" + tag]) and np.array_equal(CWD, k["cwd_" + tag])
        assert np.max(np.abs(post - k["post_" + tag])) <= 1e-4 and np.array_equal(post, k["post_" + tag])
        assert (ites == int(k["n_ite"])).all()
        rx.set_ldpc_params(int(k["n_ite"]), float(k["alpha"]), True)
        V, CWD, _, ites = rx.decode_siho(k["llr"], with_post=True)
        assert np.array_equal(np.packbits(V.astype(np.uint8), axis=1), k["bits_" + tag + "_es"]) and np.array_equal(CWD, k["cwd_" + tag + "_es"])
        assert np.array_equal(ites, k["ites_" + tag + "_es"])
    assert k["cwd_qc"].tolist() == [0, 1]
    rx.close()


def test_chain_normal_golden(Rx):
    """BASELINE configs[3]'s chain (16APSK, N = 64800 8/9, NMS 20 iterations) on one committed PL frame: demapper LLRs and information bits."""
    k = np.load(os.path.join(GOLD, "kat_chain_16apsk_normal_20ite.npz"))
    rx = Rx("16APSK-N_8/9", max_frames=1, n_ite=int(k["n_ite"]), alpha=float(k["alpha"]), early_stop=False)
    out, c0, c1 = rx.rx_bb(k["pl"], sigma=k["sigma"])
    assert np.array_equal(np.packbits(out[0].astype(np.uint8)), k["out"]) and np.array_equal(k["out"], k["info"]) and c0[0] == 1 and c1[0] == 1
    x = rx.remove_plh(rx.pl_descramble(k["pl"]))
    llr = rx.demodulate(k["sigma"], x, deinterleave=True)
    assert np.all(np.abs(llr[0] - k["llr"]) <= 1e-4 * np.maximum(1.0, np.abs(k["llr"])))
    rx.close()


def test_bch_golden(Rx):
    k = np.load(os.path.join(GOLD, "kat_bch_short.npz"))
    rx = Rx("QPSK-S_8/9", max_frames=6)
    x = np.unpackbits(k["rx"], axis=1)[:, :rx.K_ldpc].astype(np.int32)
    V, CWD = rx.decode_hiho(x)
    assert np.array_equal(np.packbits(V.astype(np.uint8), axis=1), k["out"]) and np.array_equal(CWD, k["cwd"])
    rx.close()


def test_chain_golden(Rx):
    k = np.load(os.path.join(GOLD, "kat_chain_16apsk_short.npz"))
    rx = Rx("16APSK-S_8/9", max_frames=1, n_ite=int(k["n_ite"]), alpha=float(k["alpha"]), early_stop=True)
    out, c0, c1 = rx.rx_bb(k["pl"], sigma=k["sigma"])
    assert np.array_equal(np.packbits(out[0].astype(np.uint8)), k["out"]) and c0[0] == 1 and c1[0] == 1
    x = rx.remove_plh(rx.pl_descramble(k["pl"]))
    llr = rx.demodulate(k["sigma"], x, deinterleave=True)
    assert np.all(np.abs(llr[0] - k["llr"]) <= 1e-4 * np.maximum(1.0, np.abs(k["llr"])))
    rx.close()


def test_fir_golden(Rx):
    k = np.load(os.path.join(GOLD, "kat_fir_rrc81.npz"))
    rx = Rx("QPSK-S_8/9", max_frames=1, fir_taps=k["taps"])
    assert np.max(np.abs(rx.filter(k["x1"], 1) - k["y1"])) <= 1e-4
    assert np.max(np.abs(rx.filter(k["x2"], 1) - k["y2"])) <= 1e-4
    rx.close()


def test_sync_frame_golden(Rx):
    k = np.load(os.path.join(GOLD, "kat_sync_frame_32apsk.npz"))
    rx = Rx("32APSK-S_3/4", max_frames=3)
    rx.sync_frame_set_params(float(k["alpha"]), float(k["trigger"]), int(k["vec_width"]))
    for f0 in (0, 3):
        DEL, FLG, TRI, Y = rx.sync_frame_synchronize(k["stream"][f0:f0 + 3], with_flags=True)
        assert np.array_equal(DEL, k["DEL"][f0:f0 + 3]) and np.array_equal(FLG, k["FLG"][f0:f0 + 3])
        assert np.all(np.abs(TRI - k["TRI"][f0:f0 + 3]) <= 2e-4 * np.maximum(1.0, k["TRI"][f0:f0 + 3]))
        assert np.array_equal(Y, k["Y"][f0:f0 + 3])
    rx.close()


@pytest.mark.parametrize("modcod,K,ebn0", [("QPSK-S_8/9", 14232, 4.5), ("8PSK-S_3/5", 9552, 4.5)])
def test_reference_fixed_payload_roundtrip(Rx, O, modcod, K, ebn0):
    """conf/src/K_14232.src and K_9552.src (the reference's Source_user patterns, one per code rate: DVBS2.cpp:336-349) through oracle TX -> GPU RX."""
    from helpers import chain, sigma_for
    info = np.unpackbits(np.load(os.path.join(GOLD, "src_K_%d.npy" % K)))[:K].astype(np.int32)
    ch = chain(O, modcod)
    plf, _ = ch.tx(info)
    sigma = sigma_for(ch.mc, ebn0)
    noisy = plf + (sigma * np.random.default_rng(9).standard_normal(plf.size)).astype(np.float32)
    rx = Rx(modcod, max_frames=1, n_ite=10, alpha=1.0, early_stop=True)
    out, c0, c1 = rx.rx_bb(noisy)
    assert np.array_equal(out[0], info) and c0[0] == 1 and c1[0] == 1
    rx.close()
