"""N1 parity: the device TX mirror (source .. PL scramble) vs the oracle TX chain, bit-exact
for given payloads with the channel off; noise statistics of the on-device AWGN; TX -> RX loop."""
import numpy as np
import pytest

from helpers import chain

pytestmark = pytest.mark.gpu
ALL = ["QPSK-S_8/9", "QPSK-S_3/5", "8PSK-S_3/5", "8PSK-S_8/9", "16APSK-S_8/9", "32APSK-S_3/4", "QPSK-N_8/9", "8PSK-N_8/9", "16APSK-N_8/9"]


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("modcod", ALL)
def test_tx_matches_oracle(O, Rx, modcod):
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(71)
    F = 3
    info = rng.integers(0, 2, (F, mc.K_bch)).astype(np.int32)
    info[0] = 0
    info[1] = 1
    rx = Rx(modcod, max_frames=F)
    sent, pl = rx.tx_bb(F, info=info)
    assert np.array_equal(sent, info)
    for f in range(F):
        plo, _ = ch.tx(info[f])
        assert np.array_equal(pl[f], plo), "frame %d differs at %d floats" % (f, int((pl[f] != plo).sum()))
    rx.close()


def test_tx_random_source_awgn_and_loopback(O, Rx):
    modcod = "QPSK-S_8/9"
    ch = chain(O, modcod)
    F = 64
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    s1, p1 = rx.tx_bb(F, seed=5)
    s2, p2 = rx.tx_bb(F, seed=5)
    s3, _ = rx.tx_bb(F, seed=6)
    assert np.array_equal(s1, s2) and np.array_equal(p1, p2)          # reproducible
    assert not np.array_equal(s1, s3) and not np.array_equal(s1[0], s1[1])
    assert abs(s1.mean() - 0.5) < 0.01                                 # fair bits
    # noiseless device TX equals oracle TX of the same payload
    plo, _ = ch.tx(s1[3])
    assert np.array_equal(p1[3], plo)
    # AWGN: zero mean, variance sigma^2 per real dimension, independent of the signal
    sigma = 0.4
    _, pn = rx.tx_bb(F, seed=5, sigma=sigma)
    n = (pn - p1).astype(np.float64)
    assert abs(n.mean()) < 2e-3 and abs(n.std() - sigma) < 2e-3
    assert abs(np.corrcoef(n.ravel()[:-1], n.ravel()[1:])[0, 1]) < 5e-3
    assert abs((n ** 4).mean() / n.var() ** 2 - 3.0) < 0.05           # Gaussian kurtosis
    # loopback at a workable SNR
    sent, pl = rx.tx_bb(F, seed=9, sigma=0.30)
    out, c0, c1 = rx.rx_bb(pl)
    assert np.array_equal(out, sent) and (c1 == 1).all()
    rx.close()
