"""a3/a4/a6/a7/a8/a9 parity vs the CPU oracle.  Soft values within 1e-4 (stated per test);
index maps and integer work bit-exact."""
import numpy as np
import pytest

from helpers import chain, make_pl_frames

pytestmark = pytest.mark.gpu
ALL = ["QPSK-S_8/9", "QPSK-S_3/5", "8PSK-S_3/5", "8PSK-S_8/9", "16APSK-S_8/9", "32APSK-S_3/4", "QPSK-N_8/9", "8PSK-N_8/9", "16APSK-N_8/9"]
EBN0 = {"QPSK-S_8/9": 4.0, "8PSK-S_3/5": 3.2, "8PSK-S_8/9": 6.8, "16APSK-S_8/9": 7.6, "32APSK-S_3/4": 9.0,
        "QPSK-N_8/9": 4.0, "16APSK-N_8/9": 7.6, "QPSK-S_3/5": 1.6, "8PSK-N_8/9": 6.8}


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("modcod", ALL)
def test_front_stages_match_oracle(O, Rx, modcod):
    ch = chain(O, modcod)
    mc = ch.mc
    F = 2
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, EBN0[modcod], seed=31)
    rx = Rx(modcod, max_frames=F)
    # a7
    d = rx.pl_descramble(pl)
    do = np.stack([O.pl_scramble(pl[f], 90, False) for f in range(F)])
    assert np.array_equal(d, do)
    x = rx.remove_plh(d)
    xo = np.stack([O.framer_remove_plh(do[f], mc.N_xfec) for f in range(F)])
    assert np.array_equal(x, xo)
    # a6: the M2M4 estimator subtracts nearly equal moments (2 M2^2 - M4), so the fp32 summation
    # ORDER shows up at ~1e-5 relative in sigma: the reference sums serially
    # (Estimator_DVBS2.hxx:36-41), the GPU sums per lane then by tree.  Tolerance 1e-4 relative
    # against the oracle, and the GPU must be at least as close to the fp64 value as the oracle.
    sig, eb, es = rx.estimate(x)
    for f in range(F):
        e = O.estimate(xo[f], mc.code_rate, mc.bps)
        assert abs(sig[f] - e[0]) <= 1e-4 * abs(e[0])
        assert abs(eb[f] - e[1]) <= 2e-3 and abs(es[f] - e[2]) <= 2e-3      # dB
        p = (xo[f].astype(np.float64).reshape(-1, 2) ** 2).sum(axis=1)
        m2, m4 = p.mean(), (p * p).mean()
        se = np.sqrt(abs(2 * m2 * m2 - m4))
        s64 = np.sqrt(1.0 / (2.0 * se / abs(m2 - se)))
        assert abs(sig[f] - s64) <= abs(e[0] - s64) + 2e-6 * s64
    # a3: |LLR error| <= 1e-4 * max(1, |LLR|)   (north_star: 1e-4 LLR tolerance)
    llr = rx.demodulate(np.full(F, sigma, np.float32), x)
    llro = np.stack([O.demodulate(ch.cstl, mc.bps, np.float32(sigma), xo[f]) for f in range(F)])
    assert np.all(np.abs(llr - llro) <= 1e-4 * np.maximum(1.0, np.abs(llro)))
    # a4: exact permutation
    nat = rx.deinterleave(llro)
    nato = np.empty_like(llro)
    nato[:, ch.lut] = llro
    assert np.array_equal(nat, nato)
    fused = rx.demodulate(np.full(F, sigma, np.float32), x, deinterleave=True)
    assert np.array_equal(fused[:, ch.lut], llr)
    # hard decisions of the demapper agree with the transmitted codeword almost everywhere
    hd = (nat < 0).astype(np.int32)
    assert (hd != cw).mean() < 0.2
    rx.close()


def test_bb_descramble_and_monitor(O, Rx):
    modcod = "QPSK-S_8/9"
    mc = chain(O, modcod).mc
    rng = np.random.default_rng(2)
    F = 5
    U = rng.integers(0, 2, (F, mc.K_bch)).astype(np.int32)
    rx = Rx(modcod, max_frames=F)
    S = rx.bb_descramble(U)
    So = np.stack([O.bb_scramble(U[f]) for f in range(F)])
    assert np.array_equal(S, So)
    assert np.array_equal(rx.bb_descramble(S), U)          # involution
    V = U.copy()
    V[1, :7] ^= 1
    V[3, 100] ^= 1
    rx.check_errors(U, V)
    assert rx.monitor_get() == (5, 8, 2)
    rx.check_errors(U, U)
    assert rx.monitor_get() == (10, 8, 2)
    rx.monitor_reset()
    assert rx.monitor_get() == (0, 0, 0)
    rx.close()


def test_monitor_check_errors2_sockets(O, Rx):
    """Monitor_BFER::check_errors2 (bound RX/main_sched.cpp:222-223, sockets read by probes :244-247): the counters after
    every frame of the call, BER / FER with Monitor_BFER's 1 / FRA bound before the first error."""
    modcod = "QPSK-S_8/9"
    K = chain(O, modcod).mc.K_bch
    rng = np.random.default_rng(3)
    F = 7
    U = rng.integers(0, 2, (F, K)).astype(np.int32)
    V = U.copy()
    V[2, :5] ^= 1
    V[3, 9] ^= 1
    V[6, 1000:1003] ^= 1
    rx = Rx(modcod, max_frames=F)
    fra, be, fe, ber, fer = rx.check_errors2(U, V)
    e = np.array([0, 0, 5, 1, 0, 0, 3])
    cb, cf, n = np.cumsum(e), np.cumsum(e > 0), np.arange(1, F + 1)
    assert fra.dtype == np.int64 and be.dtype == np.int32 and fer.dtype == np.float32
    assert np.array_equal(fra, n) and np.array_equal(be, cb) and np.array_equal(fe, cf)
    want_fer = np.where(cb > 0, cf / n, 1.0 / n).astype(np.float32)
    want_ber = np.where(cb > 0, cb / n / K, 1.0 / n / K).astype(np.float32)
    assert np.allclose(fer, want_fer, rtol=1e-6) and np.allclose(ber, want_ber, rtol=1e-6)
    assert rx.monitor_get() == (7, 9, 3)
    fra, be, fe, ber, fer = rx.check_errors2(U[:2], U[:2])          # the counters carry on across calls and tasks
    assert fra.tolist() == [8, 9] and be.tolist() == [9, 9] and fe.tolist() == [3, 3]
    rx.check_errors(U, V)
    assert rx.monitor_get() == (16, 18, 6)
    assert rx.monitor_reduce() == (16, 18, 6)                       # no communicator on this handle: the local counters
    rx.close()


@pytest.mark.parametrize("modcod", ["16APSK-S_8/9", "32APSK-S_3/4", "8PSK-S_8/9", "8PSK-N_8/9"])
def test_sigma_per_frame_and_high_snr_signs(O, Rx, modcod):
    """CP socket holds one sigma per frame; at high SNR the LLR sign is the hard decision.  With sigma = 0.05 the LLRs run into the hundreds: the general demapper's reference
    exponent k |y|^2 (round 4: no running maximum over the points) and its per-subset form for symbols whose weaker subset underflows are both on the path, for every
    constellation that takes the general demapper."""
    ch = chain(O, modcod)
    mc = ch.mc
    F = 3
    info, pl, cw, _ = make_pl_frames(O, modcod, F, 30.0, seed=1)
    rx = Rx(modcod, max_frames=F)
    x = rx.remove_plh(rx.pl_descramble(pl))
    sig = np.array([0.05, 0.1, 0.2], np.float32)
    nat = rx.demodulate(sig, x, deinterleave=True)
    assert np.array_equal((nat < 0).astype(np.int32), cw)
    for f in range(F):
        o = O.demodulate(ch.cstl, mc.bps, sig[f], x[f])
        nato = np.empty_like(o)
        nato[ch.lut] = o
        assert np.all(np.abs(nat[f] - nato) <= 1e-4 * np.maximum(1.0, np.abs(nato)))
    rx.close()


@pytest.mark.parametrize("modcod", ["QPSK-S_8/9", "QPSK-N_8/9"])
def test_qpsk_general_demapper_equals_the_linear_form(O, Rx, modcod, monkeypatch):
    """The reference's QPSK mapping is separable (one bit per axis), for which the library evaluates the exact LLR in its
    linear closed form.  DVBS2HIP_DEMAP_GENERAL=1 (read at create) keeps the general log-sum-exp demapper for 2-bit
    constellations: both must agree with the oracle's pairwise max* form, stand-alone and inside the fused front end."""
    ch = chain(O, modcod)
    mc = ch.mc
    F = 2
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, 4.0, seed=77)
    res = {}
    for general in (False, True):
        if general:
            monkeypatch.setenv("DVBS2HIP_DEMAP_GENERAL", "1")
        else:
            monkeypatch.delenv("DVBS2HIP_DEMAP_GENERAL", raising=False)
        rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
        x = rx.remove_plh(rx.pl_descramble(pl))
        llr = rx.demodulate(np.full(F, sigma, np.float32), x)
        llro = np.stack([O.demodulate(ch.cstl, mc.bps, np.float32(sigma), x[f]) for f in range(F)])
        assert np.all(np.abs(llr - llro) <= 1e-4 * np.maximum(1.0, np.abs(llro)))
        out, c0, c1 = rx.rx_bb(pl)                                   # estimated sigma: the register-resident fused kernel
        assert np.array_equal(out, info)
        res[general] = llr
        rx.close()
    assert np.all(np.abs(res[True] - res[False]) <= 1e-4 * np.maximum(1.0, np.abs(res[False])))


@pytest.mark.parametrize("n_cplx,energy,F", [(8370, 1.0, 5), (2 * 8370, 0.5, 3), (3402, 1.0, 64), (33282, 1.0, 2), (7, 2.0, 4), (18000, 1.0, 2), (20480, 1.0, 2), (20481, 1.0, 2)])
def test_agc_matches_oracle(O, Rx, n_cplx, energy, F):
    """Multiplier_AGC_cc_naive::imultiply (the reference's `front_agc` on 2 pl_frame osf values at energy 1 / osf, `mult_agc` on 2 pl_frame values at energy 1; RX/main_sched.cpp:197,205):
    every frame over its own standard deviation.  The reference adds its floats in order, the kernel sums in double over a fixed tree: the gain agrees with the oracle's float
    order to 1e-4 and with the sums in double to 2e-6, and a frame's result does not depend on its neighbours."""
    rng = np.random.default_rng(n_cplx)
    x = (rng.standard_normal((F, 2 * n_cplx)) * rng.uniform(0.05, 30.0, (F, 1)) + rng.uniform(-0.2, 0.2, (F, 1))).astype(np.float32)
    rx = Rx("QPSK-S_8/9", max_frames=64)
    z = rx.agc(x, n_frames=F, output_energy=energy).reshape(F, 2 * n_cplx)
    for f in range(F):
        zo = O.agc(x[f], energy)
        c = x[f, 0::2].astype(np.float64) + 1j * x[f, 1::2].astype(np.float64)
        std = np.sqrt(np.mean(np.abs(c) ** 2) - np.abs(np.mean(c)) ** 2) / np.sqrt(energy)
        assert np.max(np.abs(z[f] - zo)) <= 1e-4 * np.max(np.abs(zo))
        assert np.max(np.abs(z[f] - x[f] / std)) <= 2e-6 * np.max(np.abs(x[f] / std))
        zc = z[f, 0::2].astype(np.float64) + 1j * z[f, 1::2]
        assert abs(np.mean(np.abs(zc - zc.mean()) ** 2) - energy) < 1e-5 * energy
    alone = rx.agc(x[1], n_frames=1, output_energy=energy)
    assert np.array_equal(alone, z[1])                                       # bit for bit whatever the batch
    # device-resident form, in place
    import torch
    d = torch.from_numpy(x).cuda()
    rx.agc_dev(d.data_ptr(), d.data_ptr(), n_cplx, energy, F)
    rx.synchronize()
    assert np.array_equal(d.cpu().numpy(), z)
    with pytest.raises(Exception):
        rx.agc(x, n_frames=F, output_energy=0.0)
    rx.close()


def test_coarse_frequency_shift_matches_oracle_across_calls_and_the_counter_wrap(O, Rx):
    """Synchronizer_freq_coarse::synchronize in the transmission phase = Multiplier_sine_ccc_naive::imultiply (Synchronizer_freq_coarse_DVBS2_aib.cpp:43-50): the stream times
    exp(j omega n) with omega = 2 pi floor(-f 1e6) / 1e6 in float and n the stream position, which starts over after 999999.  Against the oracle's sample-by-sample restatement
    (same float phase, libm's cosf / sinf): 2e-6 of the largest sample; the position carries from call to call and across the wrap; FRQ / PHS report the frequency and 0."""
    rng = np.random.default_rng(5)
    n_cplx, F = 33480, 8                                                    # QPSK-S at osf 2: 267840 samples per call, the fourth call crosses n = 1e6
    f_est = 0.0123456789
    rx = Rx("QPSK-S_8/9", max_frames=F)
    rx.sync_coarse_set_freq(f_est)
    pos = 0.0
    for call in range(5):
        x = rng.standard_normal(2 * n_cplx * F).astype(np.float32)
        FRQ, PHS, y = rx.sync_coarse_synchronize(x, n_frames=F)
        yo, pos = O.nco(x, -f_est, pos)
        assert np.max(np.abs(y - yo)) <= 2e-6 * np.max(np.abs(x)), call
        assert np.all(FRQ == np.float32(-np.floor(np.float32(-f_est) * np.float32(1e6)) / np.float32(1e6))) and not PHS.any()
    assert pos == (5 * n_cplx * F) % 1000000
    # a shifted stream comes back: the channel's shift by +f, then the synchronizer set to f
    rx.sync_coarse_reset()
    x = rng.standard_normal(2 * n_cplx * F).astype(np.float32)
    f = 0.001234
    t = np.arange(n_cplx * F, dtype=np.float64)
    c = (x[0::2] + 1j * x[1::2]) * np.exp(2j * np.pi * f * t)
    sh = np.empty_like(x); sh[0::2] = c.real; sh[1::2] = c.imag
    rx.sync_coarse_set_freq(f)
    _, _, back = rx.sync_coarse_synchronize(sh, n_frames=F)
    assert np.max(np.abs(back - x)) < 2e-3 * np.max(np.abs(x))              # (float phase omega n at n ~ 2.7e5: 1e-4 rad)
    with pytest.raises(Exception):
        rx.sync_coarse_set_freq(0.7)
    rx.close()
