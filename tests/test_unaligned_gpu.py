"""Device sockets that are only 8-byte aligned (a view two floats into an allocation): the 16-byte forms of the streaming kernels
(QPSK front end, frame synchronizer's matrix-core correlators and delay line, fine synchronizers' rotation, monitor) must fall back to
their narrower forms and give the same sockets as on 16-byte aligned buffers."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_pl_frames

pytestmark = pytest.mark.gpu
vp = C.c_void_p


def _dev_pair(torch, host, shift):
    """the array on the device, `shift` elements into a fresh allocation"""
    flat = torch.empty(host.size + 8, dtype=torch.from_numpy(host.ravel()[:1]).dtype, device="cuda")
    view = flat[shift:shift + host.size]
    view.copy_(torch.from_numpy(np.ascontiguousarray(host).ravel()))
    return flat, view


@pytest.mark.parametrize("modcod", ["QPSK-S_8/9", "QPSK-N_8/9"])
def test_rx_bb_and_monitor_on_8_byte_aligned_sockets(O, modcod):
    import torch
    from dvbs2_amd.receiver import Dvbs2Hip
    F = 6
    info, pl, cw, sigma = make_pl_frames(O, modcod, F, 4.3, seed=99)
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    res = []
    for shift in (0, 2):                                     # 0: 16-byte aligned, 2: eight bytes further
        keep, d_pl = _dev_pair(torch, pl, shift)
        keep2, d_out = _dev_pair(torch, np.zeros((F, rx.K_bch), np.int32), shift)
        assert (d_pl.data_ptr() % 16 == 0) == (shift == 0)
        c0 = torch.zeros(F, dtype=torch.int8, device="cuda"); c1 = torch.zeros_like(c0)
        rx.rx_bb_dev(d_pl.data_ptr(), None, d_out.data_ptr(), c0.data_ptr(), c1.data_ptr(), F)
        rx.synchronize()
        res.append((d_out.cpu().numpy().reshape(F, -1), c0.cpu().numpy(), c1.cpu().numpy()))
        # the monitor on the same pair of sockets: U = what was sent, one int further for the second round
        keep3, d_u = _dev_pair(torch, info.astype(np.int32), shift // 2)
        rx.monitor_reset()
        rx.check_errors_dev(d_u.data_ptr(), d_out.data_ptr(), F)
        assert rx.monitor_get() == (F, int((res[-1][0] != info).sum()), int((res[-1][0] != info).any(axis=1).sum()))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][0], info)
    rx.close()


def test_synchronizers_on_8_byte_aligned_sockets(O):
    import torch
    from dvbs2_amd.receiver import Dvbs2Hip
    modcod, F, off = "QPSK-S_8/9", 14, 555
    _, pl, _, _ = make_pl_frames(O, modcod, F, 9.0, seed=5)
    n = pl.shape[1] // 2
    stream = np.concatenate([np.zeros(2 * off, np.float32), pl.reshape(-1)])[:F * 2 * n].reshape(F, 2 * n)
    out = []
    for shift in (0, 2):
        rx = Dvbs2Hip(modcod, max_frames=F)
        keep, d_x = _dev_pair(torch, stream, shift)
        keep2, d_y = _dev_pair(torch, np.zeros_like(stream), shift)
        DEL = torch.zeros(F, dtype=torch.int32, device="cuda"); FLG = torch.zeros_like(DEL)
        TRI = torch.zeros(F, dtype=torch.float32, device="cuda"); FRQ = torch.zeros_like(TRI); PHS = torch.zeros_like(TRI)
        rx.sync_frame_synchronize_dev(vp(d_x.data_ptr()), vp(DEL.data_ptr()), vp(FLG.data_ptr()), vp(TRI.data_ptr()), vp(d_y.data_ptr()), F)
        rx.synchronize()
        y_sync = d_y.cpu().numpy().copy()
        rx._chk(rx.L.dvbs2hip_sync_lr_synchronize_dev(rx.h, vp(d_x.data_ptr()), vp(FRQ.data_ptr()), vp(PHS.data_ptr()), vp(d_y.data_ptr()), F))
        rx.synchronize()
        out.append((DEL.cpu().numpy(), FLG.cpu().numpy(), TRI.cpu().numpy(), y_sync, FRQ.cpu().numpy(), d_y.cpu().numpy().copy()))
        rx.close()
    a, b = out
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[0][-1] == off      # delays, flags
    assert np.allclose(a[2], b[2], rtol=2e-4)                # metric: matrix cores (aligned) against the vector kernel (not aligned)
    assert np.array_equal(a[3], b[3])                        # aligned frames: copies
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])      # L&R: the two-sample rotation does the same arithmetic per sample
