"""Workload for the rocprofv3 passes over the NON-LDPC kernels (tools/profile_kernels.sh): every stage of the RX inner path and
the N4 synchronizers a few times each, at the sizes of BASELINE's configs.  GPU box only; prints nothing the profiler needs."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dvbs2_amd.receiver import Dvbs2Hip
from dvbs2_amd import params as P

dev = torch.device("cuda", 0)
REPS = 4
vp = ctypes.c_void_p


def chain(modcod, F, n_ite, ebn0):
    mc = P.get_modcod(modcod)
    rx = Dvbs2Hip(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=False)
    sigma = P.esn0_to_sigma(P.ebn0_to_esn0(ebn0, mc.code_rate, mc.bps))
    pl = torch.empty((F, 2 * rx.pl_frame), dtype=torch.float32, device=dev)
    sent = torch.empty((F, rx.K_bch), dtype=torch.int32, device=dev); got = torch.empty_like(sent)
    sig = torch.full((F,), sigma, dtype=torch.float32, device=dev)
    rx.tx_bb_dev(None, 1, sig.data_ptr(), sent.data_ptr(), pl.data_ptr(), F); rx.synchronize()
    for _ in range(REPS):
        rx.rx_bb_dev(pl.data_ptr(), sig.data_ptr() if mc.bps >= 4 else None, got.data_ptr(), None, None, F)
    rx.synchronize()
    # stand-alone BCH task on int32 sockets (Decoder_BCH_DVBS2::decode_hiho), the received word = a codeword with 3 bit errors per frame
    # (the three uses of bch_decode_kernel get grids of their own so that the summary can tell them apart: the fused chain above runs
    # min(F, 2048) workgroups, the clean stand-alone task 1536, the one that has errors to correct 1024)
    zero = torch.zeros((1536, rx.K_ldpc), dtype=torch.int32, device=dev)
    out = torch.empty((1536, rx.K_bch), dtype=torch.int32, device=dev); cwd = torch.empty(1536, dtype=torch.int8, device=dev)
    for _ in range(REPS):
        rx.decode_hiho_dev(zero.data_ptr(), cwd.data_ptr(), out.data_ptr(), 1536)
    cwb = torch.zeros((1024, rx.K_ldpc), dtype=torch.int32, device=dev)
    cwb[:, 5] = 1; cwb[:, 77] = 1; cwb[:, 1234] = 1
    for _ in range(REPS):
        rx.decode_hiho_dev(cwb.data_ptr(), cwd.data_ptr(), out.data_ptr(), 1024)
    rx.synchronize(); rx.close()


def sync(modcod, F):
    rx = Dvbs2Hip(modcod, max_frames=F)
    n = rx.pl_frame
    mc = P.get_modcod(modcod)
    Ft = min(F, 512)                      # a real PL stream starting mid-frame: the synchronizer locks, as in service
    tx = Dvbs2Hip(modcod, max_frames=Ft)
    sig = torch.full((Ft,), P.esn0_to_sigma(P.ebn0_to_esn0(10.0, mc.code_rate, mc.bps)), dtype=torch.float32, device=dev)
    plt = torch.empty((Ft, 2 * n), dtype=torch.float32, device=dev); sent = torch.empty((Ft, tx.K_bch), dtype=torch.int32, device=dev)
    tx.tx_bb_dev(None, 3, sig.data_ptr(), sent.data_ptr(), plt.data_ptr(), Ft); tx.synchronize(); tx.close()
    flat = plt.reshape(-1).repeat(-(-(F + 1) // Ft))
    x = flat[2468: 2468 + F * 2 * n].reshape(F, 2 * n).contiguous(); y = torch.empty_like(x)
    DEL = torch.empty(F, dtype=torch.int32, device=dev); FLG = torch.empty_like(DEL); TRI = torch.empty(F, dtype=torch.float32, device=dev)
    FRQ = torch.empty(F, dtype=torch.float32, device=dev); PHS = torch.empty_like(FRQ)
    for _ in range(REPS):
        rx.sync_frame_synchronize_dev(vp(x.data_ptr()), vp(DEL.data_ptr()), vp(FLG.data_ptr()), vp(TRI.data_ptr()), vp(y.data_ptr()), F)
        rx._chk(rx.L.dvbs2hip_sync_lr_synchronize_dev(rx.h, vp(x.data_ptr()), vp(FRQ.data_ptr()), vp(PHS.data_ptr()), vp(y.data_ptr()), F))
        rx._chk(rx.L.dvbs2hip_sync_freq_phase_synchronize_dev(rx.h, vp(x.data_ptr()), vp(FRQ.data_ptr()), vp(PHS.data_ptr()), vp(y.data_ptr()), F))
    rx.synchronize(); rx.close()


def fir(n_cplx, F, n_in, F_in):
    """a5 matched filter on a stream of F frames of n_cplx samples; N2 shaping filter on F_in frames of n_in symbols (osf 2)"""
    rx = Dvbs2Hip("32APSK-S_3/4", max_frames=max(F, F_in))
    x = torch.randn((F, 2 * n_cplx), dtype=torch.float32, device=dev); y = torch.empty_like(x)
    xs = torch.randn((F_in, 2 * n_in), dtype=torch.float32, device=dev); ys = torch.empty((F_in, 4 * n_in), dtype=torch.float32, device=dev)
    for _ in range(REPS):
        rx.filter_dev(x.data_ptr(), y.data_ptr(), n_cplx, F)
        rx.shape_filter_dev(xs.data_ptr(), ys.data_ptr(), n_in, F_in)
    rx.synchronize(); rx.close()


if "sync" not in sys.argv[1:]:             # `pmc_workload.py sync`: the synchronizers only (quick kernel-trace passes)
    fir(66564, 1024, 33282, 1024)
    fir(6804, 4096, 3402, 4096)
    chain("8PSK-N_8/9", 4096, 10, 7.5)
    chain("QPSK-N_8/9", 4096, 10, 4.2)
    chain("QPSK-S_8/9", 8192, 10, 4.4)
    chain("16APSK-N_8/9", 4096, 20, 8.2)
    chain("32APSK-S_3/4", 4096, 10, 10.5)
sync("QPSK-N_8/9", 1024)
sync("32APSK-S_3/4", 4096)
print("pmc workload done")
