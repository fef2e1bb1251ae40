"""a2 parity: HIP BCH decoder vs the CPU oracle -- integer work, bit-exact, including the
failure cases (more than t errors) and the CWD status."""
import numpy as np
import pytest

from helpers import chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("modcod", ["QPSK-S_8/9", "QPSK-S_3/5", "QPSK-N_8/9", "32APSK-S_3/4"])
def test_bch_matches_oracle(O, Rx, modcod):
    ch = chain(O, modcod)
    mc = ch.mc
    t = mc.bch_t
    rng = np.random.default_rng(21)
    n_err = [0, 1, 2, t - 1, t, t, t + 1, t + 2, 2 * t, 40, 0, 3]
    F = len(n_err)
    info = rng.integers(0, 2, (F, mc.K_bch)).astype(np.int32)
    cw = ch.bch.encode(info)
    rx_in = cw.copy()
    for f, ne in enumerate(n_err):
        pos = rng.choice(mc.N_bch, ne, replace=False)
        rx_in[f, pos] ^= 1
    # errors confined to the parity part / first and last positions
    rx_in[10, [0, mc.N_bch - 1, mc.K_bch, mc.K_bch - 1]] ^= 1
    rx = Rx(modcod, max_frames=F)
    V, CWD = rx.decode_hiho(rx_in)
    Vo, cwdo = ch.bch.decode(rx_in)
    assert np.array_equal(V, Vo)
    assert np.array_equal(CWD, cwdo)
    ok = [f for f, ne in enumerate(n_err) if ne <= t]
    assert np.array_equal(V[ok], info[ok]) and (CWD[ok] == 1).all()
    assert CWD[n_err.index(t + 1)] == 0 or not np.array_equal(V[n_err.index(t + 1)], info[n_err.index(t + 1)])
    rx.close()


def test_bch_random_garbage_matches_oracle(O, Rx):
    """Inputs that are not near any codeword (what a failed LDPC frame looks like)."""
    modcod = "QPSK-S_8/9"
    ch = chain(O, modcod)
    rng = np.random.default_rng(4)
    F = 16
    x = rng.integers(0, 2, (F, ch.mc.N_bch)).astype(np.int32)
    x[0] = 0
    x[1] = 1
    rx = Rx(modcod, max_frames=F)
    V, CWD = rx.decode_hiho(x)
    Vo, cwdo = ch.bch.decode(x)
    assert np.array_equal(V, Vo) and np.array_equal(CWD, cwdo)
    rx.close()


def test_bch_encode_decode_roundtrip_large_batch(O, Rx):
    modcod = "QPSK-N_8/9"
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(8)
    base = ch.bch.encode(rng.integers(0, 2, (4, mc.K_bch)).astype(np.int32))
    F = 600
    idx = rng.integers(0, 4, F)
    x = base[idx].copy()
    for f in range(F):
        ne = int(rng.integers(0, mc.bch_t + 1))
        x[f, rng.choice(mc.N_bch, ne, replace=False)] ^= 1
    rx = Rx(modcod, max_frames=F)
    V, CWD = rx.decode_hiho(x)
    assert np.array_equal(V, base[idx][:, :mc.K_bch]) and (CWD == 1).all()
    rx.close()
