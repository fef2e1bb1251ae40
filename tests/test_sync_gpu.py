"""Row N4 parity on the GPU: frame synchronizer vs the CPU oracle's restatement of
Synchronizer_frame_DVBS2_fast (correlations within 1e-4 of unit-power signals, the delay exact, the
aligned output bit-exact since it only copies samples)."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from helpers import make_pl_frames

pytestmark = pytest.mark.gpu
TOL = 2e-4        # sums of up to 64 products of unit-power samples, accumulated in a different order


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("modcod,batch", [("QPSK-S_8/9", 1), ("QPSK-S_8/9", 3), ("16APSK-S_8/9", 2), ("32APSK-S_3/4", 4)])
def test_synchronize_matches_oracle_frame_by_frame(O, Rx, modcod, batch):
    F, off = 12, 777
    _, pl, _, _ = make_pl_frames(O, modcod, F, 8.0, seed=4)
    n = pl.shape[1] // 2
    stream = np.concatenate([np.zeros(2 * off, np.float32), pl.reshape(-1)])[:F * 2 * n].reshape(F, 2 * n)
    sf = O.SyncFrame(n, alpha=0.9, trigger=30.0, vec_width=8)
    rx = Rx(modcod, max_frames=batch)
    for f0 in range(0, F, batch):
        x = stream[f0:f0 + batch]
        d, flg, tri, Y = rx.sync_frame_synchronize(x, with_flags=True)
        for k in range(batch):
            do, Yo = sf.synchronize(x[k])
            assert d[k] == do, (f0 + k, d[k], do)
            assert np.array_equal(Y[k], Yo), f0 + k
            assert abs(tri[k] - sf.metric) <= TOL * max(1.0, sf.metric) and bool(flg[k]) == sf.packet_flag     # TRI / FLG sockets
        m, flag = rx.sync_frame_metric()
        assert abs(m - sf.metric) <= TOL * max(1.0, sf.metric) and flag == sf.packet_flag
    assert d[-1] == off
    rx.close()


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
@pytest.mark.parametrize("modcod,F", [("QPSK-N_8/9", 5), ("32APSK-S_3/4", 37), ("QPSK-S_8/9", 1)])
def test_fused_one_task_form_equals_the_two_task_path(O, Rx, monkeypatch, modcod, F, kernel):
    """synchronize (one task) keeps the two correlations on chip (sync_corr_m_kernel: several blocks per frame, halo of cor_SOF
    recomputed, first block from the handle's history); DVBS2HIP_SYNC_UNFUSED sends it through synchronize1 + synchronize2 over
    device scratch.  Same delays, flags, metric BITS and aligned frames, across calls (the histories carry over) -- with the
    correlators on the matrix cores (k_sync_mfma.hip, the default) and as fp32 vector sums (DVBS2HIP_SYNC=valu)."""
    if kernel == "valu":
        monkeypatch.setenv("DVBS2HIP_SYNC", "valu")
    rng = np.random.default_rng(12)
    _, pl, _, _ = make_pl_frames(O, modcod, min(F, 6), 9.0, seed=7)
    n = pl.shape[1] // 2
    reps = -(-F // pl.shape[0])
    base = np.concatenate([np.zeros(2 * 333, np.float32), np.tile(pl.reshape(-1), reps)])
    a, b = Rx(modcod, max_frames=F), Rx(modcod, max_frames=F)
    for call in range(3):
        x = (base[call * 17 * 2: call * 17 * 2 + F * 2 * n] + 0.05 * rng.standard_normal(F * 2 * n).astype(np.float32)).reshape(F, 2 * n)
        d1, f1, t1, Y1 = a.sync_frame_synchronize(x, with_flags=True)
        monkeypatch.setenv("DVBS2HIP_SYNC_UNFUSED", "1")
        d2, f2, t2, Y2 = b.sync_frame_synchronize(x, with_flags=True)
        monkeypatch.delenv("DVBS2HIP_SYNC_UNFUSED")
        assert np.array_equal(d1, d2) and np.array_equal(f1, f2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32)), call
        assert np.array_equal(Y1, Y2), call
    a.close(); b.close()


@pytest.mark.parametrize("modcod,F", [("QPSK-N_8/9", 3), ("32APSK-S_3/4", 5)])
def test_matrix_core_correlators_against_the_vector_kernel_and_fp64(O, Rx, monkeypatch, modcod, F):
    """The +-1 taps are exact in bf16 and the differential samples are split exactly into three bf16 terms, so the matrix-core sums
    differ from the vector kernel's (the reference's order of fp32 additions) by rounding order only: both within 4e-6 of the fp64
    correlation on unit-power input, the delays and the aligned frames identical, over two calls (the memories carry over)."""
    rng = np.random.default_rng(5)
    _, pl, _, _ = make_pl_frames(O, modcod, F, 5.0, seed=9)
    n = pl.shape[1] // 2
    stream = np.concatenate([0.7 * rng.standard_normal(2 * 4321).astype(np.float32), pl.reshape(-1), pl.reshape(-1)])[:2 * F * 2 * n]
    a, b = Rx(modcod, max_frames=F), Rx(modcod, max_frames=F)
    z = np.concatenate([np.zeros(63), [1.0], stream[0::2].astype(np.float64) + 1j * stream[1::2]])       # reg_channel = (1, 0), .cpp:19
    dz = np.zeros(len(z), complex)
    dz[1:] = z[:-1] * np.conj(z[1:])
    sof = np.array([1, -1, -1, 1, -1, 1, 1, -1, 1, 1, -1, -1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, 1, 1, 1], float)
    ref_sof = np.convolve(dz, sof)[64:64 + 2 * F * n]
    for call in range(2):
        x = stream[call * F * 2 * n:(call + 1) * F * 2 * n].reshape(F, 2 * n)
        monkeypatch.delenv("DVBS2HIP_SYNC", raising=False)                     # (whatever the environment of the test run says)
        cs1, cp1 = a.sync_frame_synchronize1(x)
        d1, Y1 = a.sync_frame_synchronize2(x, cs1, cp1)
        monkeypatch.setenv("DVBS2HIP_SYNC", "valu")
        cs2, cp2 = b.sync_frame_synchronize1(x)
        d2, Y2 = b.sync_frame_synchronize2(x, cs2, cp2)
        monkeypatch.delenv("DVBS2HIP_SYNC", raising=False)
        r = ref_sof[call * F * n:(call + 1) * F * n]
        for cs in (cs1, cs2):
            got = cs.reshape(-1)[0::2].astype(np.float64) + 1j * cs.reshape(-1)[1::2]
            assert np.max(np.abs(got - r)) <= 4e-6 * 25, call
        assert np.max(np.abs(cs1 - cs2)) <= 4e-6 * 25 and np.max(np.abs(cp1 - cp2)) <= 4e-6 * 32, call
        assert not np.array_equal(cp1, cp2)                                    # two different kernels did run
        assert np.array_equal(d1, d2) and np.array_equal(Y1, Y2), call
    a.close(); b.close()


def test_frame_synchronizer_at_size_locks_and_realigns_the_stream(O, Rx):
    """BASELINE-size call (611 normal frames = 20 M samples in ONE call: many workgroups per frame in the correlators, 25 whole chunks of 24
    frames + a partial one in the average, the delay line in lock) through size-independent properties: the delay settles on the stream's
    offset, the flag is up, the metric is the same for every period of the (6-periodic) stream once the average has converged, and every
    output frame is, bit for bit, the PL frame that started `off` samples into the previous input frame."""
    modcod, F, off = "QPSK-N_8/9", 611, 12345
    _, pl, _, _ = make_pl_frames(O, modcod, 6, 10.0, seed=21)
    n = pl.shape[1] // 2
    stream = np.concatenate([np.zeros(2 * off, np.float32), np.tile(pl.reshape(-1), F // 6 + 1)])[:F * 2 * n].reshape(F, 2 * n)
    rx = Rx(modcod, max_frames=F)
    d, flg, tri, Y = rx.sync_frame_synchronize(stream, with_flags=True)
    assert np.all(d[8:] == off) and np.all(flg[8:] == 1)                 # (the average over frames needs a few of them)
    for f in range(9, F):
        assert np.array_equal(Y[f], pl[(f - 1) % 6]), f
    assert np.array_equal(tri[200:206], tri[206:212]) and np.all(tri[200:] > 30.0)
    # the same stream cut into calls of 100, 1 and 510 frames: same sockets (the memories carry over)
    rx2 = Rx(modcod, max_frames=F)
    parts = [rx2.sync_frame_synchronize(stream[a:b], with_flags=True) for a, b in ((0, 100), (100, 101), (101, F))]
    assert np.array_equal(np.concatenate([q[0] for q in parts]), d) and np.array_equal(np.concatenate([q[3] for q in parts]), Y)
    assert np.array_equal(np.concatenate([q[2] for q in parts]).view(np.uint32), tri.view(np.uint32))
    rx.close(); rx2.close()


def test_average_and_arg_max_over_many_short_frames(O, Rx):
    """The thirteen-wave form of the average over frames / arg max (k_sync.hip, sync_metric_argmax_kernel<96>: short frames, 96 frames between two barriers) on a call of
    2 whole chunks + a partial one, then calls of 96, 1 and 132 frames (a whole chunk exactly, less than one, one + a partial one): delays and aligned frames equal to the
    oracle's frame by frame, the metric within the correlators' tolerance, and every socket bit for bit what round 3's four-wave kernel gives (a child process with
    DVBS2HIP_SYNC_ARGMAX4 set: the knob is read once per process)."""
    import subprocess, sys, tempfile
    modcod, F, off = "32APSK-S_3/4", 229, 1501
    _, pl, _, _ = make_pl_frames(O, modcod, 7, 9.0, seed=31)
    n = pl.shape[1] // 2
    stream = np.concatenate([np.zeros(2 * off, np.float32), np.tile(pl.reshape(-1), F // 7 + 2)])[:F * 2 * n].reshape(F, 2 * n)
    sf = O.SyncFrame(n, alpha=0.9, trigger=30.0, vec_width=8)
    rx = Rx(modcod, max_frames=F)
    d, flg, tri, Y = rx.sync_frame_synchronize(stream, with_flags=True)
    for f in range(F):
        do, Yo = sf.synchronize(stream[f])
        assert d[f] == do and np.array_equal(Y[f], Yo), f
        assert abs(tri[f] - sf.metric) <= TOL * max(1.0, sf.metric) and bool(flg[f]) == sf.packet_flag, f
    assert d[-1] == off
    rx2 = Rx(modcod, max_frames=F)
    parts = [rx2.sync_frame_synchronize(stream[a:b], with_flags=True) for a, b in ((0, 96), (96, 97), (97, F))]
    assert np.array_equal(np.concatenate([q[0] for q in parts]), d) and np.array_equal(np.concatenate([q[3] for q in parts]), Y)
    assert np.array_equal(np.concatenate([q[2] for q in parts]).view(np.uint32), tri.view(np.uint32))
    rx.close(); rx2.close()
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "x.npy"), stream)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from dvbs2_amd.receiver import Dvbs2Hip; x = np.load(%r); rx = Dvbs2Hip(%r, max_frames=%d); "
                "d, flg, tri, Y = rx.sync_frame_synchronize(x, with_flags=True); np.savez(%r, d=d, flg=flg, tri=tri, Y=Y)"
                % (ROOT, os.path.join(td, "x.npy"), modcod, F, os.path.join(td, "o.npz")))
        subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, DVBS2HIP_SYNC_ARGMAX4="1"))
        o = np.load(os.path.join(td, "o.npz"))
        assert np.array_equal(o["d"], d) and np.array_equal(o["flg"], flg) and np.array_equal(o["Y"], Y)
        assert np.array_equal(o["tri"].view(np.uint32), tri.view(np.uint32))


def test_two_task_form_and_correlations(O, Rx):
    modcod = "8PSK-S_3/5"
    F = 5
    _, pl, _, _ = make_pl_frames(O, modcod, F, 6.0, seed=5)
    n = pl.shape[1] // 2
    sf = O.SyncFrame(n)
    rx = Rx(modcod, max_frames=F)
    # synchronize1 on the whole batch at once = the stream form of the two correlators; the reference runs
    # the two tasks alternately per frame (synchronize2 is what moves reg_channel on, .cpp:294)
    cs, cp = rx.sync_frame_synchronize1(pl)
    d, Y = rx.sync_frame_synchronize2(pl, cs, cp)
    for f in range(F):
        c1, c2 = sf.synchronize1(pl[f])
        do, Yo = sf.synchronize2(pl[f], c1, c2)
        assert np.max(np.abs(cs[f] - c1)) <= TOL * 25 and np.max(np.abs(cp[f] - c2)) <= TOL * 32
        assert d[f] == do and np.array_equal(Y[f], Yo)
    rx.close()


def test_reset_parameters_and_delay_changes(O, Rx):
    modcod = "QPSK-S_8/9"
    _, pl, _, _ = make_pl_frames(O, modcod, 10, 9.0, seed=6)
    n = pl.shape[1] // 2
    flat = pl.reshape(-1)
    rx = Rx(modcod, max_frames=2)
    rx.sync_frame_set_params(alpha=0.5, trigger=10.0, vec_width=16)
    sf = O.SyncFrame(n, alpha=0.5, trigger=10.0, vec_width=16)
    # the offset of the stream changes twice: the delay line goes through its transitional branches
    offs = [100, 100, 100, 4000, 4000, 4000, 50, 50]
    pos = 0
    for f, off in enumerate(offs):
        x = np.roll(flat, 2 * off)[pos:pos + 2 * n].copy()
        pos += 2 * n
        d, Y = rx.sync_frame_synchronize(x)
        do, Yo = sf.synchronize(x)
        assert d[0] == do and np.array_equal(Y[0], Yo), f
    m, flag = rx.sync_frame_metric()
    assert flag == sf.packet_flag
    rx.sync_frame_reset(); sf.reset()
    x = flat[:2 * n]
    d, Y = rx.sync_frame_synchronize(x)
    do, Yo = sf.synchronize(x)
    assert d[0] == do and np.array_equal(Y[0], Yo)
    rx.close()


def _rot(O, pl_frame, f0, ph):
    d = O.pl_scramble(pl_frame, scramble=False)
    n = d.size // 2
    c = (d[0::2] + 1j * d[1::2]) * np.exp(2j * np.pi * (f0 * np.arange(n) + ph))
    x = np.empty(2 * n, np.float32)
    x[0::2], x[1::2] = c.real, c.imag
    return x


@pytest.mark.parametrize("modcod", ["QPSK-S_8/9", "QPSK-N_8/9", "16APSK-S_8/9"])
def test_fine_synchronizers_match_oracle(O, Rx, modcod):
    """Synchronizer_freq_phase_DVBS2_aib and Synchronizer_Luise_Reggiannini_DVBS2_aib: estimates within 1e-6 absolute
    (float sums in the reference's order; atan2f may differ by an ulp), rotated frames within 2e-4 (cos / sin of a phase
    of up to a few hundred radians held in fp32)."""
    F = 4
    _, pl, _, _ = make_pl_frames(O, modcod, F, 10.0, seed=9)
    n = pl.shape[1] // 2
    x = np.stack([_rot(O, pl[f], 1e-4 * (f + 1), 0.1 * f) for f in range(F)])
    rx = Rx(modcod, max_frames=F)
    FRQ, PHS, Y = rx.sync_freq_phase_synchronize(x)
    for f in range(F):
        fo, po, Yo = O.sync_freq_phase(x[f])
        assert abs(FRQ[f] - fo) <= 1e-6 and abs(PHS[f] - po) <= 1e-4
        assert np.max(np.abs(Y[f] - Yo)) <= 4e-3        # a 1e-6 difference in frequency is 2 pi n 1e-6 rad at the frame's end
    rx.sync_lr_set_alpha(0.7)
    lr = O.SyncLR(n, alpha=0.7)
    for call in range(2):                               # the damped autocorrelation is carried across calls
        FRQ, PHS, Y = rx.sync_lr_synchronize(x)
        for f in range(F):
            fo, po, Yo = lr.synchronize(x[f])
            assert abs(FRQ[f] - fo) <= 1e-6 and PHS[f] == 0.0
            assert np.max(np.abs(Y[f] - Yo)) <= 4e-3
    rx.sync_lr_reset(); lr.reset()
    FRQ, _, _ = rx.sync_lr_synchronize(x[:1])
    assert abs(FRQ[0] - lr.synchronize(x[0])[0]) <= 1e-6
    rx.close()


@pytest.mark.parametrize("modcod,F", [("32APSK-S_3/4", 131), ("QPSK-S_8/9", 64), ("QPSK-N_8/9", 3), ("32APSK-S_3/4", 1500)])
def test_lr_recurrence_and_rotation_in_one_launch_equal_the_three_kernel_path(O, Rx, monkeypatch, modcod, F):
    """The L&R synchronizer's default form runs the recurrence over the frames in one workgroup of the launch that rotates the frames (k_sync.hip, sff_lr_fused_kernel: the
    rotating workgroups wait for their frame's estimate); DVBS2HIP_LR=unfused is the recurrence and the rotation as kernels of their own.  Same values in the same order:
    estimates and rotated frames bit for bit, over two calls (the damped autocorrelation and the launch's counters are carried), batches that are not whole multiples of 64
    frames, and against the oracle on the first and the last frames."""
    _, pl, _, _ = make_pl_frames(O, modcod, min(F, 6), 10.0, seed=11)
    n = pl.shape[1] // 2
    base = [_rot(O, pl[f], 1e-4 * (f + 1), 0.1 * f) for f in range(pl.shape[0])]
    rng = np.random.default_rng(5)
    x = np.stack([base[f % len(base)] for f in range(F)]) + (0.05 * rng.standard_normal((F, 2 * n))).astype(np.float32)
    out = {}
    for form in ("fused", "unfused"):
        if form == "unfused":
            monkeypatch.setenv("DVBS2HIP_LR", "unfused")
        else:
            monkeypatch.delenv("DVBS2HIP_LR", raising=False)
        rx = Rx(modcod, max_frames=F)
        rx.sync_lr_set_alpha(0.9)
        out[form] = [rx.sync_lr_synchronize(x), rx.sync_lr_synchronize(x[::-1].copy()), rx.sync_lr_synchronize(x[:max(1, F // 3)])]
        rx.close()
    for a, b in zip(out["fused"], out["unfused"]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    lr = O.SyncLR(n, alpha=0.9)
    FRQ, _, Y = out["fused"][0]
    for f in range(F):
        fo, _, Yo = lr.synchronize(x[f])
        if f < 3 or f >= F - 3:
            assert abs(FRQ[f] - fo) <= 1e-6 and np.max(np.abs(Y[f] - Yo)) <= 4e-3


@pytest.mark.parametrize("modcod,F", [("32APSK-S_3/4", 700), ("QPSK-N_8/9", 5)])
def test_lr_waiting_workgroup_that_gives_up_is_reported_and_the_call_repaired(O, Rx, monkeypatch, modcod, F):
    """a rotating workgroup of sff_lr_fused_kernel that does not see its frame's estimate in time no longer traps (which took the whole context down): it
    drops its stores and sets the handle's error word; the host form rotates the call again before it returns, the device form when dvbs2hip_synchronize is called.
    DVBS2HIP_LR_TIMEOUT_US=0 makes every workgroup give up at its first unsuccessful poll: the outputs must still equal the three-kernel path's, bit for bit,
    dvbs2hip_sync_lr_timeouts must count the repeats, and the handle must keep working with the normal limit afterwards."""
    import torch
    _, pl, _, _ = make_pl_frames(O, modcod, min(F, 4), 10.0, seed=12)
    n = pl.shape[1] // 2
    rng = np.random.default_rng(6)
    x = np.stack([_rot(O, pl[f % pl.shape[0]], 2e-4 * (f % 7 + 1), 0.05 * f) for f in range(F)]) + (0.05 * rng.standard_normal((F, 2 * n))).astype(np.float32)
    monkeypatch.setenv("DVBS2HIP_LR", "unfused")
    rx = Rx(modcod, max_frames=F); rx.sync_lr_set_alpha(0.9)
    ref = [rx.sync_lr_synchronize(x), rx.sync_lr_synchronize(x[::-1].copy())]
    rx.close()
    monkeypatch.delenv("DVBS2HIP_LR")
    monkeypatch.setenv("DVBS2HIP_LR_TIMEOUT_US", "0")
    rx = Rx(modcod, max_frames=F); rx.sync_lr_set_alpha(0.9)
    got = [rx.sync_lr_synchronize(x)]
    n_host = rx.sync_lr_timeouts()
    # device form: nothing is looked at before dvbs2hip_synchronize
    xd = torch.from_numpy(x[::-1].copy()).cuda(); yd = torch.empty_like(xd); fd = torch.empty(F, dtype=torch.float32, device="cuda"); pd = torch.empty_like(fd)
    torch.cuda.synchronize()
    rx._chk(rx.L.dvbs2hip_sync_lr_synchronize_dev(rx.h, xd.data_ptr(), fd.data_ptr(), pd.data_ptr(), yd.data_ptr(), F))
    rx.synchronize()
    got.append((fd.cpu().numpy(), pd.cpu().numpy(), yd.cpu().numpy().reshape(F, -1)))
    n_dev = rx.sync_lr_timeouts()
    for a, b in zip(got, ref):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2].reshape(F, -1), b[2].reshape(F, -1))
    assert n_dev >= n_host and (F < 8 or n_dev >= 1)          # (a handful of frames may all be published before the first poll)
    monkeypatch.delenv("DVBS2HIP_LR_TIMEOUT_US")
    again = rx.sync_lr_synchronize(x)
    assert rx.sync_lr_timeouts() == n_dev and again[2].shape == ref[0][2].shape
    rx.close()


def test_lr_timeouts_of_several_outstanding_device_calls_are_each_repaired_with_their_own_estimates(O, Rx, monkeypatch):
    """the repair of a device-form L&R call that timed out needs THAT launch's estimates and THAT launch's error word.  Sequences that used to
    corrupt Y silently (one error word, one `lr_last`, estimates in a buffer shared with the next call): lr_dev(A) -> lr_dev(B) -> freq_phase_dev(C) -> a host-form L&R call
    -> dvbs2hip_synchronize, every L&R launch forced to time out.  Every output must equal the three-kernel path's (state carried call to call), bit for bit."""
    import torch
    modcod, F = "QPSK-S_8/9", 48
    _, pl, _, _ = make_pl_frames(O, modcod, 4, 10.0, seed=14)
    n = pl.shape[1] // 2
    rng = np.random.default_rng(8)
    xs = [np.stack([_rot(O, pl[(f + k) % pl.shape[0]], 2e-4 * ((f + k) % 7 + 1), 0.05 * f + 0.3 * k) for f in range(F)])
          + (0.05 * rng.standard_normal((F, 2 * n))).astype(np.float32) for k in range(4)]
    monkeypatch.setenv("DVBS2HIP_LR", "unfused")
    rx = Rx(modcod, max_frames=F); rx.sync_lr_set_alpha(0.9)
    ref = [rx.sync_lr_synchronize(xs[0]), rx.sync_lr_synchronize(xs[1]), rx.sync_freq_phase_synchronize(xs[2]), rx.sync_lr_synchronize(xs[3])]
    rx.close()
    monkeypatch.delenv("DVBS2HIP_LR")
    monkeypatch.setenv("DVBS2HIP_LR_TIMEOUT_US", "0")
    rx = Rx(modcod, max_frames=F); rx.sync_lr_set_alpha(0.9)
    dev = []
    for k in range(3):
        xd = torch.from_numpy(xs[k]).cuda(); yd = torch.empty_like(xd); fd = torch.empty(F, dtype=torch.float32, device="cuda"); pd = torch.empty_like(fd)
        dev.append((xd, yd, fd, pd))
    torch.cuda.synchronize()
    for k in range(3):
        xd, yd, fd, pd = dev[k]
        fn = rx.L.dvbs2hip_sync_lr_synchronize_dev if k < 2 else rx.L.dvbs2hip_sync_freq_phase_synchronize_dev
        rx._chk(fn(rx.h, xd.data_ptr(), fd.data_ptr(), pd.data_ptr(), yd.data_ptr(), F))
    host = rx.sync_lr_synchronize(xs[3])           # a host-form call in between: must not swallow the device calls' error words
    rx.synchronize()
    assert rx.sync_lr_timeouts() >= 2
    for k in range(3):
        _, yd, fd, _ = dev[k]
        assert np.array_equal(fd.cpu().numpy(), ref[k][0]), k
        assert np.array_equal(yd.cpu().numpy().reshape(F, -1), ref[k][2].reshape(F, -1)), k
    assert np.array_equal(host[0], ref[3][0]) and np.array_equal(host[2].reshape(F, -1), ref[3][2].reshape(F, -1))
    rx.close()


def test_lr_one_launch_form_under_another_handles_persistent_kernel():
    """The rotating workgroups of sff_lr_fused_kernel wait for words that workgroup 0 publishes, which is safe because workgroup 0 is placed first.  Issued while another
    handle's persistent LDPC launch owns every CU (tools/lr_soak.py), the synchronizer's workgroups are placed a few at a time as LDPC workgroups retire: every call must
    still end (a workgroup that waits for a second gives up and the host repeats the rotation: none does) with the three-kernel path's output."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lr_soak.py"), "25"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "225 calls, 0 bad" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("modcod,F,ebn0", [("QPSK-S_8/9", 24, 6.0), ("8PSK-S_8/9", 10, 9.0), ("32APSK-S_3/4", 17, 14.0), ("QPSK-N_8/9", 6, 6.0)])
def test_located_form_feeds_the_fused_chain_what_the_delayed_copy_would(O, Rx, modcod, F, ebn0):
    """VERDICT r4 item 5, the chained device form: dvbs2hip_sync_frame_locate_dev returns WHERE every aligned frame starts (inside the input stream when the frame is one run
    of it, inside scratch otherwise) and dvbs2hip_rx_bb_located_dev reads the frames there -- against dvbs2hip_sync_frame_synchronize_dev + dvbs2hip_rx_bb_dev on the delayed
    copy: same DEL / FLG / TRI, same information bits and CWD flags, frame for frame, over four calls on a stream that starts mid-frame (the delay moves while the synchronizer
    acquires: those frames are materialized; in lock every frame but the first and last of a call is read in place) and whose offset JUMPS in the third call.  Two handles
    run the two forms side by side: the synchronizer's state (delay line, last output frame) carries from call to call in both.  QPSK (pair kernel, frames on 8-byte
    addresses), 8PSK (frame in registers), 32APSK (two-sweep kernel), QPSK normal frames."""
    import torch
    info, pl, _, sigma = make_pl_frames(O, modcod, min(F, 6), ebn0, seed=31)
    n = pl.shape[1] // 2
    reps = -(-4 * F // pl.shape[0]) + 2
    flat = np.tile(pl.reshape(-1), reps)
    # calls 0, 1: offset 1501 symbols; from call 2 on the stream has slipped by 77 symbols more (a new alignment: the delay moves again)
    base = np.concatenate([np.zeros(2 * 1501, np.float32), flat])
    slipped = np.concatenate([np.zeros(2 * (1501 + 77), np.float32), flat])
    a, b = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True), Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    dev = torch.device("cuda")
    K = a.K_bch
    sg = torch.full((F,), float(sigma), dtype=torch.float32, device=dev) if a.bps >= 4 else None
    in_place = 0
    for call in range(4):
        src_stream = base if call < 2 else slipped
        x = torch.from_numpy(src_stream[call * F * 2 * n:(call + 1) * F * 2 * n].copy()).to(dev)
        out = {}
        for form, rx in (("copy", a), ("located", b)):
            DEL = torch.empty(F, dtype=torch.int32, device=dev); FLG = torch.empty_like(DEL); TRI = torch.empty(F, dtype=torch.float32, device=dev)
            bits = torch.empty((F, K), dtype=torch.int32, device=dev); c0 = torch.empty(F, dtype=torch.int8, device=dev); c1 = torch.empty_like(c0)
            torch.cuda.synchronize()
            if form == "copy":
                Y = torch.empty_like(x)
                rx.sync_frame_synchronize_dev(x.data_ptr(), DEL.data_ptr(), FLG.data_ptr(), TRI.data_ptr(), Y.data_ptr(), F)
                rx.rx_bb_dev(Y.data_ptr(), sg.data_ptr() if sg is not None else None, bits.data_ptr(), c0.data_ptr(), c1.data_ptr(), F)
            else:
                SRC = torch.zeros(F, dtype=torch.int64, device=dev)
                rx.sync_frame_locate_dev(x.data_ptr(), DEL.data_ptr(), FLG.data_ptr(), TRI.data_ptr(), SRC.data_ptr(), F)
                rx.rx_bb_located_dev(SRC.data_ptr(), sg.data_ptr() if sg is not None else None, bits.data_ptr(), c0.data_ptr(), c1.data_ptr(), F)
            rx.synchronize()
            out[form] = (DEL.cpu().numpy(), FLG.cpu().numpy(), TRI.cpu().numpy(), bits.cpu().numpy(), c0.cpu().numpy(), c1.cpu().numpy())
            if form == "located":
                p = SRC.cpu().numpy()
                inside = (p >= x.data_ptr()) & (p < x.data_ptr() + x.numel() * 4)
                in_place += int(inside.sum())
                assert (p % 8 == 0).all() and not inside[0] and not inside[-1]           # the first and the last frame of a call are materialized
        for u, v in zip(out["copy"], out["located"]):
            assert np.array_equal(u, v), call
        if call in (1, 3):      # in lock: the payloads come out, and nearly every frame was read in place
            assert (out["located"][1][2:] == 1).all() and out["located"][5][3:].all()
    assert in_place >= 2 * (F - 2) - 4, in_place
    a.close(); b.close()
