"""Row N4, CPU: the oracle's restatement of Synchronizer_frame_DVBS2_fast and Variable_delay_cc_naive
(in-tree reference code, followed line by line) does what a frame synchronizer has to do on a real PL
frame stream, and its constants are the reference's."""
import numpy as np

from helpers import make_pl_frames


def test_correlator_taps_are_the_reference_header(O):
    # Synchronizer_frame_DVBS2_fast.hpp:19-33: conj_SOF = differential SOF (25 taps), conj_PLSC zero at odd positions
    sof, plsc = O.sync_frame_taps()
    assert sof.size == 25 and plsc.size == 64
    assert set(np.unique(sof)) == {-1.0, 1.0} and np.all(plsc[1::2] == 0) and set(np.unique(plsc[0::2])) == {-1.0, 1.0}
    # the differential of the SOF the framer transmits correlates fully with conj_SOF: build it from the PLHEADER
    plh = O.plheader([0, 0, 1, 0, 1, 0, 1])          # any PLS code: the first 26 symbols are the SOF
    c = plh[0:52:2] + 1j * plh[1:52:2]
    d = c[:-1] * np.conj(c[1:])                       # the reference's differential signal, .cpp:138-142
    # y[i] = sum_m b[m] d[i-m]: at the last SOF symbol the correlator sees d[24-m] under tap m
    corr = np.sum(sof * d[::-1])
    assert abs(abs(corr) - 25.0) < 1e-4


def test_variable_delay_is_a_delay_and_follows_the_reference_when_it_changes(O):
    N = 40
    vd = O.VariableDelay(N, 5, 20)
    rng = np.random.default_rng(1)
    x = [rng.standard_normal(N).astype(np.float32) for _ in range(4)]
    y0 = vd.filter(x[0])
    assert np.all(y0[:10] == 0) and np.array_equal(y0[10:], x[0][:30])            # first call: zeros, then the input
    y1 = vd.filter(x[1])
    assert np.array_equal(y1[:10], x[0][30:]) and np.array_equal(y1[10:], x[1][:30])   # steady state: a pure delay
    vd.set_delay(8)                                                                # the delay grows by 3 samples
    y2 = vd.filter(x[2])
    # Variable_delay_cc_naive.cpp:62-73: start_Y = 16 - 10 = 6 floats re-read from the previous output's tail,
    # then the buffered tail of the previous input, then the new input from float 16 on
    assert np.array_equal(y2[:6], y1[N - 6:]) and np.array_equal(y2[6:16], x[1][30:]) and np.array_equal(y2[16:], x[2][:24])
    vd.set_delay(100)                                                              # clamped to max_delay (.cpp:91-95)
    y3 = vd.filter(x[3])
    assert y3.size == N and np.array_equal(y3[40:], x[3][:0])


def test_frame_sync_finds_the_offset_and_aligns_the_stream(O):
    modcod = "QPSK-S_8/9"
    F, off = 7, 1234
    _, pl, _, _ = make_pl_frames(O, modcod, F, 6.0, seed=3)
    n = pl.shape[1] // 2
    stream = np.concatenate([np.zeros(2 * off, np.float32), pl.reshape(-1)])
    sf = O.SyncFrame(n, alpha=0.9, trigger=30.0, vec_width=8)
    delays, metrics = [], []
    for f in range(F):
        d, Y = sf.synchronize(stream[f * 2 * n:(f + 1) * 2 * n])
        delays.append(d); metrics.append(sf.metric)
        if f >= 3:
            assert d == off
            assert np.array_equal(Y, pl[f - 1])          # the output is the previous PL frame, aligned to its SOF
    assert delays[-1] == off and metrics[-1] > metrics[2]
    # synchronize == synchronize1 + synchronize2 (the two-task form of the pipelined mains)
    sa, sb = O.SyncFrame(n), O.SyncFrame(n)
    for f in range(3):
        x = stream[f * 2 * n:(f + 1) * 2 * n]
        da, Ya = sa.synchronize(x)
        cs, cp = sb.synchronize1(x)
        db, Yb = sb.synchronize2(x, cs, cp)
        assert da == db and np.array_equal(Ya, Yb)
    # reset (.cpp:304-318): detection starts over, the metric falls back
    sa.reset()
    sa.synchronize(stream[0:2 * n])
    assert sa.metric < metrics[-1]


def test_frame_sync_tail_is_not_averaged(O):
    # .cpp:284-285: positions past the last full SIMD vector get the instantaneous value
    n = 8370
    rng = np.random.default_rng(2)
    x = rng.standard_normal(2 * n).astype(np.float32)
    a, b = O.SyncFrame(n, alpha=0.9, vec_width=8), O.SyncFrame(n, alpha=0.0, vec_width=8)
    a.synchronize(x); b.synchronize(x)
    x2 = rng.standard_normal(2 * n).astype(np.float32)
    # with one random frame of history both see the same instantaneous correlation in the 2-sample tail
    cs_a, cp_a = a.synchronize1(x2)
    cs_b, cp_b = b.synchronize1(x2)
    assert np.array_equal(cs_a, cs_b) and n % 8 == 2


def _rotated(O, pl_frame, f0, ph):
    d = O.pl_scramble(pl_frame, scramble=False)          # the fine synchronizers run after Scrambler_PL::descramble
    n = d.size // 2
    c = (d[0::2] + 1j * d[1::2]) * np.exp(2j * np.pi * (f0 * np.arange(n) + ph))
    x = np.empty(2 * n, np.float32)
    x[0::2], x[1::2] = c.real, c.imag
    return d, x


def test_fine_frequency_and_phase_synchronizers(O):
    modcod = "QPSK-S_8/9"
    _, pl, _, _ = make_pl_frames(O, modcod, 3, 14.0, seed=8)
    n = pl.shape[1] // 2
    assert list(O.sff_pilots(n)) == [1530, 3006, 4482, 5958, 7434]      # 90 + 16 slots, then every 16 slots + 36
    f0, ph = 1.5e-4, -0.2
    d, x = _rotated(O, pl[0], f0, ph)
    frq, phs, Y = O.sync_freq_phase(x)
    # Synchronizer_freq_phase_DVBS2_aib: least-squares line through the unwrapped pilot phases
    assert abs(frq - f0) < 2e-5 and abs((phs - ph + 0.5) % 1.0 - 0.5) < 0.02
    err = np.abs((Y[0::2] + 1j * Y[1::2]) - (d[0::2] + 1j * d[1::2]))[90:]
    assert err.max() < 0.2
    # Luise & Reggiannini: noisy on one frame, the damping pulls it in over frames; PHS stays 0
    lr = O.SyncLR(n, alpha=0.9)
    est = []
    for k in range(30):
        _, xk = _rotated(O, pl[k % 3], f0, 0.0)
        est.append(lr.synchronize(xk)[0])
    assert abs(np.mean(est[-10:]) - f0) < 1.5e-4
    lr.reset()
    assert np.all(lr.R_l == 0)
