"""a5 parity: streaming SRRC FIR vs the CPU oracle (1e-4 absolute on unit-power signals),
state carried across calls, ragged sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("kernel", ["auto", "valu", "mfma"])
def test_fir_matches_oracle_and_keeps_state(O, Rx, P, kernel):
    """both kernels behind dvbs2hip_filter: the fp32 vector FIR and the matrix-core FIR (bf16 x 3 split operands)"""
    from dvbs2_amd import lib_binding as B
    taps = P.rrc_taps(0.2, 2, 20)
    assert taps.size == 81
    rng = np.random.default_rng(5)
    F, n = 3, 6804                       # 32APSK-S pl_frame 3402 * osf 2 (BASELINE config 5)
    rx = Rx("32APSK-S_3/4", max_frames=8)
    rx.set_filter_kernel({"auto": B.FIR_AUTO, "valu": B.FIR_VALU, "mfma": B.FIR_MFMA}[kernel])
    hist = np.zeros(2 * 80, np.float32)
    for call in range(3):
        x = rng.standard_normal(F * 2 * n).astype(np.float32)
        y = rx.filter(x, n_frames=F)
        yo = O.fir(taps, hist, x)
        assert np.max(np.abs(y - yo)) <= TOL
    rx.filter_reset()
    x = rng.standard_normal(2 * n).astype(np.float32)
    y = rx.filter(x, 1)
    yo = O.fir(taps, np.zeros(2 * 80, np.float32), x)
    assert np.max(np.abs(y - yo)) <= TOL
    rx.close()


@pytest.mark.parametrize("kernel", ["valu", "mfma"])
def test_fir_impulse_ragged_and_tiny(O, Rx, P, kernel):
    from dvbs2_amd import lib_binding as B
    taps = P.rrc_taps(0.35, 4, 5)        # 41 taps: run-time tap-count path (vector kernel), zero-padded band (matrix cores)
    rx = Rx("QPSK-S_8/9", max_frames=4, fir_taps=taps, fir_osf=4)
    rx.set_filter_kernel(B.FIR_VALU if kernel == "valu" else B.FIR_MFMA)
    x = np.zeros(2 * 300, np.float32)
    x[0] = 1.0
    y = rx.filter(x, 1)
    assert np.max(np.abs(y[0:2 * 41:2] - taps)) <= 1e-6 and np.max(np.abs(y[1::2])) == 0.0
    rx.filter_reset()
    # frames shorter than the filter memory: history must chain through several calls
    rng = np.random.default_rng(6)
    hist = np.zeros(2 * 40, np.float32)
    for n in (7, 1, 33, 2049, 5):
        x = rng.standard_normal(2 * n).astype(np.float32)
        y = rx.filter(x, 1)
        yo = O.fir(taps, hist, x)
        assert np.max(np.abs(y - yo)) <= TOL, n
    rx.close()


def test_fir_mfma_large_stream_tile_carry_and_tail(O, Rx, P):
    """several tiles per persistent workgroup (overlap carried in LDS), a ragged tail, state across calls; the two kernels
    agree with each other far inside the 1e-4 bar; an unaligned device socket falls back to the vector kernel; a filter
    longer than 81 taps refuses the matrix-core form"""
    import torch
    from dvbs2_amd import lib_binding as B
    taps = P.rrc_taps(0.2, 2, 20)
    rng = np.random.default_rng(21)
    F, n = 37, 66564                                         # 2 462 868 samples = 1202.6 tiles of 2048: two tiles per workgroup
    rx = Rx("QPSK-N_8/9", max_frames=F)
    hist = np.zeros(2 * 80, np.float32)
    x1 = rng.standard_normal(F * 2 * n).astype(np.float32)
    out = {}
    for kern in (B.FIR_MFMA, B.FIR_VALU):
        rx.filter_reset(); rx.set_filter_kernel(kern)
        ya = rx.filter(x1, n_frames=F)
        out[kern] = ya
    yo = O.fir(taps, hist, x1)
    assert np.max(np.abs(out[B.FIR_MFMA] - yo)) <= TOL and np.max(np.abs(out[B.FIR_VALU] - yo)) <= TOL
    assert np.max(np.abs(out[B.FIR_MFMA] - out[B.FIR_VALU])) <= 2e-5
    # second call continues the stream (filter memory written by the matrix-core kernel itself), odd sample count
    rx.filter_reset(); rx.set_filter_kernel(B.FIR_MFMA)
    rx.filter(x1, n_frames=F)
    m = 4099
    x3 = rng.standard_normal(2 * m).astype(np.float32)
    y3 = rx.filter(x3, 1)
    assert np.max(np.abs(y3 - O.fir(taps, x1[-160:].copy(), x3))) <= TOL
    # unaligned device sockets (8-byte offset): AUTO falls back to the vector kernel, results unchanged
    rx.filter_reset(); rx.set_filter_kernel(B.FIR_AUTO)
    dx = torch.zeros(2 * m + 2, dtype=torch.float32, device="cuda"); dy = torch.zeros_like(dx)
    dx[2:] = torch.from_numpy(x3).cuda()
    rx.filter_dev(dx.data_ptr() + 8, dy.data_ptr() + 8, m, 1); rx.synchronize()
    assert np.max(np.abs(dy[2:].cpu().numpy() - O.fir(taps, np.zeros(160, np.float32), x3))) <= TOL
    rx.close()
    long_taps = P.rrc_taps(0.2, 4, 20)                       # 161 taps
    rx = Rx("QPSK-S_8/9", max_frames=1, fir_taps=long_taps, fir_osf=4)
    with pytest.raises(Exception):
        rx.set_filter_kernel(B.FIR_MFMA)
    x = rng.standard_normal(2 * 3000).astype(np.float32)
    assert np.max(np.abs(rx.filter(x, 1) - O.fir(long_taps, np.zeros(320, np.float32), x))) <= TOL
    rx.close()


def test_rrc_matched_pair_is_nyquist(O, Rx, P):
    """TX shaping (oracle upfir) then the GPU matched filter: zero ISI at symbol spacing."""
    taps = P.rrc_taps(0.2, 2, 20)
    rng = np.random.default_rng(7)
    sym = (rng.integers(0, 2, 2 * 2000) * 2.0 - 1.0).astype(np.float32) / np.sqrt(2.0).astype(np.float32)
    tx = O.upfir(taps, 2, np.zeros(160, np.float32), sym)
    rx = Rx("QPSK-S_8/9", max_frames=1)
    y = rx.filter(tx, 1)
    d = 80                                # two group delays of 40 samples
    got = y.reshape(-1, 2)[d::2][:1900]
    want = sym.reshape(-1, 2)[:1900]
    assert np.max(np.abs(got - want)) < 2e-3
    rx.close()


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
def test_shape_filter_matches_oracle_and_keeps_state(O, Rx, P, kernel):
    """N2: polyphase up-sampling SRRC (Filter_UPFIR_ccr_naive.cpp:52-66) vs the oracle, 1e-4: the two branches on the
    matrix cores (osf = 2) and the one-lane-per-output vector kernel."""
    from dvbs2_amd import lib_binding as B
    taps = P.rrc_taps(0.2, 2, 20)
    rng = np.random.default_rng(15)
    rx = Rx("32APSK-S_3/4", max_frames=4)
    rx.set_filter_kernel(B.FIR_MFMA if kernel == "mfma" else B.FIR_VALU)
    hist = np.zeros(2 * 80, np.float32)
    for n, F in ((3402, 2), (7, 1), (1000, 3), (5000, 4)):
        x = rng.standard_normal(F * 2 * n).astype(np.float32)
        y = rx.shape_filter(x, n_frames=F, osf=2)
        yo = O.upfir(taps, 2, hist, x)
        assert y.size == 2 * x.size and np.max(np.abs(y - yo)) <= TOL
    rx.close()


def test_filtered_loop_awgn_extract(O, Rx, P):
    """TX shaping -> AWGN -> matched filter -> perfect-timing extraction gives back the symbols
    (+ noise of the requested variance): the filtered loop of BASELINE config 5."""
    import ctypes
    from dvbs2_amd import lib_binding as B
    rng = np.random.default_rng(16)
    n, F = 3402, 4
    rx = Rx("32APSK-S_3/4", max_frames=F)
    sym = (rng.integers(0, 2, F * 2 * n) * 2.0 - 1.0).astype(np.float32) * np.float32(np.sqrt(0.5))
    up = rx.shape_filter(sym, n_frames=F, osf=2)
    clean = rx.filter(up, n_frames=F).reshape(-1, 2)
    got = clean[80::2][: F * n - 40]                      # two group delays of 40 samples
    assert np.max(np.abs(got - sym.reshape(-1, 2)[: F * n - 40])) < 2e-3
    sigma = 0.25
    noisy = rx.add_noise(sigma, up, seed=3, n_frames=F)
    d = (noisy - up).astype(np.float64)
    assert abs(d.mean()) < 3e-3 and abs(d.std() - sigma) < 3e-3
    assert np.array_equal(noisy, rx.add_noise(sigma, up, seed=3, n_frames=F))          # reproducible
    assert not np.array_equal(noisy, rx.add_noise(sigma, up, seed=4, n_frames=F))
    # extract_dev through raw device buffers (dvbs2hip_malloc / memcpy helpers)
    L = rx.L
    din, dout = ctypes.c_void_p(), ctypes.c_void_p()
    mf = rx.filter(up, n_frames=F) if False else clean.astype(np.float32).ravel()
    assert L.dvbs2hip_malloc(rx.h, ctypes.byref(din), mf.nbytes) == 0 and L.dvbs2hip_malloc(rx.h, ctypes.byref(dout), mf.nbytes // 2) == 0
    assert L.dvbs2hip_memcpy_h2d(rx.h, din, mf.ctypes.data_as(ctypes.c_void_p), mf.nbytes) == 0
    rx.extract_dev(din.value, dout.value, n, 2, 80, F)
    out = np.empty(F * 2 * n, np.float32)
    assert L.dvbs2hip_memcpy_d2h(rx.h, out.ctypes.data_as(ctypes.c_void_p), dout, out.nbytes) == 0
    assert np.array_equal(out.reshape(-1, 2)[: F * n - 40], got) and not out.reshape(-1, 2)[F * n - 40:].any()
    L.dvbs2hip_free(rx.h, din); L.dvbs2hip_free(rx.h, dout)
    rx.close()


@pytest.mark.parametrize("kernel", ["auto", "valu"])
def test_filter1_filter2_split_matches_the_reference_tasks(O, Rx, P, kernel):
    """flt::tsk::filter1 / filter2 (Filter_FIR_ccr.cpp:144-294, bound RX/main_sched.cpp:199-201): filter2(X, filter1(X)) is the
    whole filter; each half agrees with the oracle's restatement of the reference's halves where both define the output; filter2 is
    a pure function of its sockets (garbage in the upper part of Y_N2h does not leak, no state is read or advanced)."""
    from dvbs2_amd import lib_binding as B
    taps = P.rrc_taps(0.2, 2, 20)
    rng = np.random.default_rng(15)
    F, n = 3, 6804
    rx = Rx("32APSK-S_3/4", max_frames=8)
    rx.set_filter_kernel({"auto": B.FIR_AUTO, "valu": B.FIR_VALU}[kernel])
    ref = Rx("32APSK-S_3/4", max_frames=8)
    ref.set_filter_kernel({"auto": B.FIR_AUTO, "valu": B.FIR_VALU}[kernel])
    split = rx.filter_split(n)
    assert split % 4 == 0 and 80 <= split <= n // 2
    end1, up = O.fir_split(2 * n, 81)
    assert end1 >= up                                         # the reference's halves overlap or meet
    hist = np.zeros(2 * 80, np.float32)
    for call in range(3):
        x = rng.standard_normal(F * 2 * n).astype(np.float32)
        y1 = rx.filter1(x, n_frames=F)
        y2 = rx.filter2(x, y1, n_frames=F)
        yw = ref.filter(x, n_frames=F)                        # the one-task filter on a twin handle with the same history
        assert np.array_equal(y2, yw)                         # bit for bit: same kernel, same stream
        h2 = hist.copy()
        o1 = O.fir1(taps, hist, x, n_frames=F)
        o2 = O.fir2(taps, x, o1, n_frames=F)
        assert np.max(np.abs(o2 - O.fir(taps, h2, x))) == 0.0           # the oracle's halves make the oracle's whole
        lo = min(2 * split, end1)
        for f in range(F):
            a = slice(f * 2 * n, f * 2 * n + lo)
            assert np.max(np.abs(y1[a] - o1[a])) <= TOL       # what both filter1s define
        assert np.max(np.abs(y2 - o2)) <= TOL
        # filter2 alone: Y_N2h arbitrary; its lower part is copied, its upper part ignored
        yh = rng.standard_normal(F * 2 * n).astype(np.float32)
        y3 = rx.filter2(x, yh, n_frames=F).reshape(F, 2 * n)
        assert np.array_equal(y3[:, :2 * split], yh.reshape(F, 2 * n)[:, :2 * split])
        assert np.array_equal(y3[:, 2 * split:], yw.reshape(F, 2 * n)[:, 2 * split:])
    with pytest.raises(Exception):
        rx.filter1(np.zeros(2 * 100, np.float32), 1)          # half a frame shorter than the filter's memory
    rx.close(); ref.close()
