"""Fused RX baseband chain (a7 -> a8) vs the oracle chain wired like TX_RX_BB/main.cpp:83-92,
plus the error behaviour of the C ABI."""
import numpy as np
import pytest

from helpers import chain, make_pl_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("modcod,ebn0", [("QPSK-S_8/9", 4.4), ("8PSK-S_3/5", 3.6), ("8PSK-S_8/9", 7.2),
                                         ("16APSK-S_8/9", 8.2), ("QPSK-S_3/5", 2.2), ("QPSK-N_8/9", 4.3)])
def test_rx_bb_perfect_sigma_matches_oracle(O, Rx, modcod, ebn0):
    ch = chain(O, modcod)
    F = 3
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, ebn0, seed=41)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=0.875, early_stop=True)
    out, c0, c1 = rx.rx_bb(pl, sigma=np.float32(sigma))
    for f in range(F):
        r = ch.rx(pl[f], sigma=np.float32(sigma), n_ite=10, alpha=0.875, sched=O.QC, early_stop=True)
        assert np.array_equal(out[f], r["info"])
        assert c0[f] == r["ldpc_cwd"] and c1[f] == r["bch_cwd"]
    assert np.array_equal(out, info)      # and the payload is recovered
    rx.close()


def test_rx_bb_estimated_sigma_recovers_payload(O, Rx):
    modcod = "QPSK-S_8/9"
    F = 8
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, 4.6, seed=43)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=0.875, early_stop=False)
    out, c0, c1 = rx.rx_bb(pl)
    assert np.array_equal(out, info) and (c0 == 1).all() and (c1 == 1).all()
    rx.close()


def test_rx_bb_low_snr_reports_failure(O, Rx):
    modcod = "QPSK-S_8/9"
    F = 4
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, 1.0, seed=44)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    out, c0, c1 = rx.rx_bb(pl, sigma=np.float32(sigma))
    assert (c0 == 0).all()
    assert (out != info).any()
    rx.close()


def test_abi_error_behaviour(O, Rx):
    from dvbs2_amd.lib_binding import Dvbs2HipError
    with pytest.raises(ValueError):                     # DVBS2.cpp:319 invalid_argument
        Rx("64QAM-S_1/2")
    rx = Rx("QPSK-S_8/9", max_frames=2)
    with pytest.raises(Dvbs2HipError) as ei:            # more frames than the socket holds
        rx.decode_siho(np.zeros((3, rx.N_ldpc), np.float32))
    assert ei.value.code == -1
    with pytest.raises(ValueError):                     # length_error: ragged socket
        rx.decode_siho(np.zeros(rx.N_ldpc + 1, np.float32))
    with pytest.raises(Dvbs2HipError):
        rx.set_ldpc_params(0)
    rx.close()
    with pytest.raises(Dvbs2HipError):
        Rx("QPSK-S_8/9", max_frames=0)
