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
                                         ("16APSK-S_8/9", 8.2), ("QPSK-S_3/5", 2.2), ("QPSK-N_8/9", 4.3), ("8PSK-N_8/9", 7.4)])
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


@pytest.mark.parametrize("modcod,ebn0,n_ite", [("16APSK-N_8/9", 8.2, 20), ("16APSK-N_8/9", 6.6, 20), ("32APSK-S_3/4", 9.0, 10), ("32APSK-S_3/4", 6.0, 10)])
def test_rx_bb_configs_3_and_4_match_oracle(O, Rx, modcod, ebn0, n_ite):
    """BASELINE configs[3] (16APSK N = 64800, NMS 20 ite) and the baseband part of configs[4] (32APSK-S_3/4) through the FUSED chain,
    frame by frame against the oracle chain: payload and both CWD flags at a clean point (payload recovered); well below the
    waterfall the failure path: the LDPC flags agree (nothing converges).  (The demapper meets the oracle to 1e-4, not bit for
    bit, so the hard decisions of a frame that does NOT converge may differ: only converged frames are compared bit by bit.  The
    decoder itself is compared on non-convergent frames in test_ldpc_headline_config_at_size_matches_oracle.)"""
    ch = chain(O, modcod)
    F = 4
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, ebn0, seed=45)
    rx = Rx(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=True)
    out, c0, c1 = rx.rx_bb(pl, sigma=np.float32(sigma))
    for f in range(F):
        r = ch.rx(pl[f], sigma=np.float32(sigma), n_ite=n_ite, alpha=1.0, sched=O.QC, early_stop=True)
        assert c0[f] == r["ldpc_cwd"], f
        if r["ldpc_cwd"]:
            assert np.array_equal(out[f], r["info"]) and c1[f] == r["bch_cwd"], f
    if ebn0 > 8.0:
        assert np.array_equal(out, info) and (c0 == 1).all() and (c1 == 1).all()
    else:
        assert (c0 == 0).all()
    rx.close()


@pytest.mark.parametrize("modcod,ebn0,n_ite", [("QPSK-S_8/9", 5.2, 2), ("QPSK-N_8/9", 4.9, 2), ("16APSK-S_8/9", 8.4, 2), ("QPSK-S_8/9", 3.0, 4)])
def test_rx_bb_bch_corrects_and_fails_behind_the_fused_output(O, Rx, monkeypatch, modcod, ebn0, n_ite):
    """In the fused chain the LDPC kernel writes the descrambled info bits itself and the BCH stage only patches what it corrects.
    Cut the LDPC decoder short (2-3 fixed iterations at a high SNR) so that frames reach the BCH stage with a few residual bit
    errors: BCH must correct them in place (LDPC CWD 0, BCH CWD 1, payload exact).  At 3.0 dB everything fails: the output is the
    uncorrected word.  Both ways the result equals the unfused chain's (BCH writing every bit: DVBS2HIP_CHAIN_UNFUSED) bit for bit."""
    F = 24 if "-N_" in modcod else 64
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, ebn0, seed=46)
    rx = Rx(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=False)
    out, c0, c1 = rx.rx_bb(pl, sigma=np.float32(sigma))
    monkeypatch.setenv("DVBS2HIP_CHAIN_UNFUSED", "1")
    out_u, c0_u, c1_u = rx.rx_bb(pl, sigma=np.float32(sigma))
    monkeypatch.delenv("DVBS2HIP_CHAIN_UNFUSED")
    assert np.array_equal(out, out_u) and np.array_equal(c0, c0_u) and np.array_equal(c1, c1_u)
    if ebn0 > 4.0:
        fixed = (c0 == 0) & (c1 == 1)
        assert fixed.sum() >= 3, (int(fixed.sum()), int(c0.sum()))                       # frames the BCH stage had to repair
        assert np.array_equal(out[c1 == 1], info[c1 == 1])
        # and against the oracle chain for the repaired frames
        ch = chain(O, modcod)
        for f in np.flatnonzero(fixed)[:3]:
            r = ch.rx(pl[f], sigma=np.float32(sigma), n_ite=n_ite, alpha=1.0, sched=O.QC, early_stop=False)
            assert r["bch_cwd"] == 1 and np.array_equal(out[f], r["info"])
    else:
        assert (c1 == 0).all() and (out != info).any(axis=1).all()
    rx.close()


def test_rx_bb_estimated_sigma_recovers_payload(O, Rx):
    modcod = "QPSK-S_8/9"
    F = 8
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, 4.6, seed=43)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=0.875, early_stop=False)
    out, c0, c1 = rx.rx_bb(pl)
    assert np.array_equal(out, info) and (c0 == 1).all() and (c1 == 1).all()
    rx.close()


@pytest.mark.parametrize("modcod,ebn0", [("QPSK-S_8/9", 3.9), ("QPSK-N_8/9", 3.7)])
def test_qpsk_front_end_pairs_equal_single_symbols(O, Rx, monkeypatch, modcod, ebn0):
    """The QPSK front end with two neighbouring symbols per lane and access (front_reg2_kernel, the default on 16-byte aligned sockets)
    against the one-symbol form (DVBS2HIP_FRONT_SINGLE=1): given sigma the LLRs are the same numbers, so the fused chain returns the
    same bits and flags frame for frame around the waterfall (converged or not); with the estimator (its sums run in another order,
    sigma differs in the last bits and min-sum with alpha = 1 is scale-invariant) the payload is recovered either way."""
    F = 24
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, ebn0, seed=321)
    res = {}
    for single in (False, True):
        if single:
            monkeypatch.setenv("DVBS2HIP_FRONT_SINGLE", "1")
        else:
            monkeypatch.delenv("DVBS2HIP_FRONT_SINGLE", raising=False)
        rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
        res[single] = rx.rx_bb(pl, sigma=sigma)
        rx.close()
    for a, b in zip(res[False], res[True]):
        assert np.array_equal(a, b)
    assert 0 < int(res[False][1].sum()) < F                             # some frames converge, some do not: both kinds were compared
    info2, pl2, _, _ = make_pl_frames(O, modcod, 4, ebn0 + 1.5, seed=322)
    monkeypatch.delenv("DVBS2HIP_FRONT_SINGLE", raising=False)
    rx = Rx(modcod, max_frames=4, n_ite=10, alpha=1.0, early_stop=True)
    out, c0, c1 = rx.rx_bb(pl2)
    assert np.array_equal(out, info2) and c0.all()
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
    # a caller-supplied BCH field below GF(2^8) is refused (the syndrome kernel splits remainders into a low byte and m - 8 high bits)
    import ctypes as C
    from dvbs2_amd import lib_binding as B
    L = B.load()
    cfg = B.Cfg()
    assert L.dvbs2hip_cfg_from_modcod(b"QPSK-S_8/9", C.byref(cfg)) == 0
    assert cfg.ldpc_implem == 2 and cfg.ldpc_n_ite == 50            # the reference's defaults: SPA, 50 iterations (DVBS2.cpp:135-138)
    cfg.bch_m = 7
    h = C.c_void_p()
    assert L.dvbs2hip_create(C.byref(cfg), C.byref(h)) == -1 and b"8 <= m" in L.dvbs2hip_last_error(None)
    # frames are the second grid dimension of several kernels: a socket of more than 65534 frames is refused at create
    assert L.dvbs2hip_cfg_from_modcod(b"QPSK-S_8/9", C.byref(cfg)) == 0
    cfg.max_frames = 65535
    assert L.dvbs2hip_create(C.byref(cfg), C.byref(h)) == -1 and b"65534" in L.dvbs2hip_last_error(None)


def test_external_stream_reset_and_two_handles(O, Rx):
    """cfg.stream: the handle enqueues on the caller's HIP stream (here torch's current stream), so
    torch.cuda.Event sees its kernels; two handles coexist on one device; reset() clears the filter
    memory and the monitor counters."""
    import torch
    modcod = "QPSK-S_8/9"
    F = 4
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, 4.6, seed=47)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        rx = Rx(modcod, max_frames=F, n_ite=10, early_stop=True, stream=s.cuda_stream)
        assert rx.stream == s.cuda_stream
        other = Rx("16APSK-S_8/9", max_frames=2)                       # a second handle, own stream
        d_pl = torch.from_numpy(pl).cuda()
        d_out = torch.empty((F, rx.K_bch), dtype=torch.int32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        rx.rx_bb_dev(d_pl.data_ptr(), None, d_out.data_ptr(), None, None, F)
        e1.record(s)
        s.synchronize()
        assert e0.elapsed_time(e1) > 0.0
        assert np.array_equal(d_out.cpu().numpy(), info)
    rx.check_errors(info, info)
    x = np.ones(2 * 100, np.float32)
    y1 = rx.filter(x, 1)
    rx.reset()
    assert rx.monitor_get() == (0, 0, 0)
    assert np.array_equal(rx.filter(x, 1), y1)                           # same output again: the filter memory was cleared
    other.close(); rx.close()


def test_pinned_host_sockets_give_the_same_results(O, Rx):
    """dvbs2hip_host_register: the chunked, overlapped host-form path (all sockets pinned) against the plain one,
    on a batch that does not divide into the chunks evenly; unregistering falls back to the plain path."""
    modcod, F = "QPSK-S_8/9", 333
    info, pl, _, sigma = make_pl_frames(O, modcod, 9, 4.6, seed=31)
    pl = np.ascontiguousarray(np.tile(pl, (37, 1)))[:F]
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    ref_info, ref_c0, ref_c1 = rx.rx_bb(pl, sigma=np.float32(sigma))
    out = (np.empty((F, rx.K_bch), np.int32), np.zeros(F, np.int8), np.zeros(F, np.int8))
    for a in (pl,) + out:
        rx.host_register(a)
    got = rx.rx_bb(pl, sigma=np.float32(sigma), out=out)
    assert np.array_equal(got[0], ref_info) and np.array_equal(got[1], ref_c0) and np.array_equal(got[2], ref_c1)
    assert np.array_equal(got[0][:9], info)
    # LDPC task alone
    llr = rx.demodulate(np.float32(sigma), rx.remove_plh(rx.pl_descramble(pl)), deinterleave=True)
    V0, C0 = rx.decode_siho(llr)
    llr = np.ascontiguousarray(llr)
    outl = (np.empty((F, rx.K_ldpc), np.int32), np.zeros(F, np.int8))
    for a in (llr,) + outl:
        rx.host_register(a)
    V1, C1 = rx.decode_siho(llr, out=outl)
    assert np.array_equal(V0, V1) and np.array_equal(C0, C1)
    # the streaming tasks take the same path when their sockets are pinned (the receiver's Python wrappers allocate their
    # outputs, so go through the ABI): PL descrambler, and the matched filter, whose memory has to chain through the chunks
    import ctypes as C
    y_ref = rx.pl_descramble(pl)
    y = np.empty_like(pl)
    rx.host_register(y)
    assert rx.L.dvbs2hip_pl_descramble(rx.h, pl.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), F) == 0
    assert np.array_equal(y, y_ref)
    rx.filter_reset()
    f_ref = rx.filter(pl, n_frames=F)
    rx.filter_reset()
    assert rx.L.dvbs2hip_filter(rx.h, pl.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), rx.pl_frame, F) == 0
    assert np.array_equal(y.reshape(f_ref.shape), f_ref)
    rx.host_unregister(llr)
    V2, C2 = rx.decode_siho(llr, out=outl)
    assert np.array_equal(V0, V2)
    with pytest.raises(Exception):
        rx.host_unregister(llr)
    rx.close()
