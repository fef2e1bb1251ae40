"""Python host side over the C ABI: one `Dvbs2Hip` object = one dvbs2hip handle = one GPU.

Method names and socket arguments mirror the reference's tasks (decode_siho, decode_hiho,
demodulate, deinterleave, filter, estimate, descramble, remove_plh, check_errors:
/root/reference src/mains/TX_RX_BB/main.cpp:83-94).  numpy arrays in/out (host sockets);
the `*_dev` methods take raw device pointers (ints, e.g. torch.Tensor.data_ptr()) and only
enqueue on the handle's stream.

Product code: nothing here imports or calls the CPU oracle, and nothing computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib_binding as B
from . import params as P


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class Dvbs2Hip:
    def __init__(self, modcod: str = "QPSK-S_8/9", max_frames: int = 1, n_ite: int = 50, alpha: float = 1.0,
                 early_stop: bool = True, device: int = 0, stream: int | None = None, fir_taps=None,
                 fir_osf: int = 2, lds_groups: int = -1, implem: str = "NMS"):
        self.L = B.load()
        self.h = None
        self.mc = P.get_modcod(modcod)            # raises ValueError like DVBS2.cpp:319
        cfg = B.Cfg()
        rc = self.L.dvbs2hip_cfg_from_modcod(self.mc.name.encode(), C.byref(cfg))
        if rc:
            raise B.Dvbs2HipError(rc, self.L.dvbs2hip_last_error(None).decode())
        cfg.max_frames = int(max_frames)
        cfg.ldpc_n_ite = int(n_ite)
        cfg.ldpc_alpha = float(alpha)
        if implem not in ("NMS", "MS", "SPA", "SPA_TANH", "SPA_EXACT"):
            raise ValueError("implem has to be NMS, MS, SPA, SPA_TANH or SPA_EXACT")
        cfg.ldpc_implem = {"NMS": 0, "MS": 1, "SPA": 2, "SPA_TANH": 3, "SPA_EXACT": 4}[implem]
        cfg.ldpc_early_stop = 1 if early_stop else 0
        cfg.device = int(device)
        cfg.stream = stream
        cfg.ldpc_lds_groups = int(lds_groups)
        self._taps = None
        if fir_taps is not None:
            self._taps = np.ascontiguousarray(fir_taps, dtype=np.float32)
            cfg.fir_n_taps = self._taps.size
            cfg.fir_taps = self._taps.ctypes.data
            cfg.fir_osf = int(fir_osf)
        h = C.c_void_p()
        rc = self.L.dvbs2hip_create(C.byref(cfg), C.byref(h))
        if rc:
            raise B.Dvbs2HipError(rc, self.L.dvbs2hip_last_error(None).decode())
        self.h = h
        self.max_frames = int(max_frames)
        sz = B.Sizes()
        self._chk(self.L.dvbs2hip_get_sizes(self.h, C.byref(sz)))
        self.N_ldpc, self.K_ldpc, self.K_bch, self.bps = sz.N_ldpc, sz.K_ldpc, sz.K_bch, sz.bps
        self.N_xfec, self.pl_frame, self.ldpc_edges, self.ldpc_q = sz.N_xfec_sym, sz.pl_frame_sym, sz.ldpc_edges, sz.ldpc_q

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc:
            raise B.Dvbs2HipError(rc, self.L.dvbs2hip_last_error(self.h).decode())

    def close(self):
        if self.h is not None:
            self.L.dvbs2hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _frames(self, a, per_frame, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.size % per_frame:
            raise ValueError("socket size %d is not a multiple of the frame size %d" % (a.size, per_frame))
        return a.reshape(-1, per_frame), a.size // per_frame

    @property
    def stream(self) -> int:
        return int(self.L.dvbs2hip_get_stream(self.h) or 0)

    def synchronize(self):
        self._chk(self.L.dvbs2hip_synchronize(self.h))

    def reset(self):
        self._chk(self.L.dvbs2hip_reset(self.h))

    def set_ldpc_params(self, n_ite, alpha=1.0, early_stop=True):
        self._chk(self.L.dvbs2hip_set_ldpc_params(self.h, int(n_ite), float(alpha), 1 if early_stop else 0))

    # ------------------------------------------------------------------ a1
    def host_register(self, arr):
        """pins a host socket buffer (numpy array) for the handle's lifetime: see dvbs2hip_host_register"""
        self._chk(self.L.dvbs2hip_host_register(self.h, _ptr(arr), arr.nbytes))

    def host_unregister(self, arr):
        self._chk(self.L.dvbs2hip_host_unregister(self.h, _ptr(arr)))

    def decode_siho(self, Y_N, with_post=False, out=None):
        """out = (V_K int32[F,K_ldpc], CWD int8[F]) reuses the caller's socket buffers (e.g. pinned ones)"""
        Y, F = self._frames(Y_N, self.N_ldpc, np.float32)
        if out is not None:
            V, CWD = out
            assert V.dtype == np.int32 and V.size == F * self.K_ldpc and CWD.dtype == np.int8 and CWD.size == F and V.flags.c_contiguous
        else:
            V = np.empty((F, self.K_ldpc), dtype=np.int32)
            CWD = np.zeros(F, dtype=np.int8)
        if not with_post:
            self._chk(self.L.dvbs2hip_ldpc_decode_siho(self.h, _ptr(Y), _ptr(CWD), _ptr(V), F))
            return V, CWD
        post = np.empty((F, self.N_ldpc), dtype=np.float32)
        ites = np.zeros(F, dtype=np.int32)
        self._chk(self.L.dvbs2hip_ldpc_decode_siho_post(self.h, _ptr(Y), _ptr(CWD), _ptr(V), _ptr(post), _ptr(ites), F))
        return V, CWD, post, ites

    def decode_siho_dev(self, Y_N, CWD, V_K, n_frames):
        self._chk(self.L.dvbs2hip_ldpc_decode_siho_dev(self.h, _ptr(Y_N), _ptr(CWD), _ptr(V_K), n_frames))

    # ------------------------------------------------------------------ a2
    def decode_hiho(self, Y_N):
        Y, F = self._frames(Y_N, self.K_ldpc, np.int32)
        V = np.empty((F, self.K_bch), dtype=np.int32)
        CWD = np.zeros(F, dtype=np.int8)
        self._chk(self.L.dvbs2hip_bch_decode_hiho(self.h, _ptr(Y), _ptr(CWD), _ptr(V), F))
        return V, CWD

    def decode_hiho_dev(self, Y_N, CWD, V_K, n_frames):
        self._chk(self.L.dvbs2hip_bch_decode_hiho_dev(self.h, _ptr(Y_N), _ptr(CWD), _ptr(V_K), n_frames))

    # ------------------------------------------------------------------ a3 / a4
    def demodulate(self, CP, Y_N1, deinterleave=False):
        Y, F = self._frames(Y_N1, 2 * self.N_xfec, np.float32)
        cp = np.ascontiguousarray(np.broadcast_to(np.asarray(CP, dtype=np.float32).ravel(), (F,)))
        out = np.empty((F, self.N_ldpc), dtype=np.float32)
        fn = self.L.dvbs2hip_demodulate_deinterleave if deinterleave else self.L.dvbs2hip_demodulate
        self._chk(fn(self.h, _ptr(cp), _ptr(Y), _ptr(out), F))
        return out

    def deinterleave(self, itl):
        X, F = self._frames(itl, self.N_ldpc, np.float32)
        out = np.empty_like(X)
        self._chk(self.L.dvbs2hip_deinterleave(self.h, _ptr(X), _ptr(out), F))
        return out

    # ------------------------------------------------------------------ a5
    def filter(self, X_N1, n_frames=1):
        X = np.ascontiguousarray(X_N1, dtype=np.float32).ravel()
        if X.size % (2 * n_frames):
            raise ValueError("filter socket size must be a multiple of 2 * n_frames")
        out = np.empty_like(X)
        self._chk(self.L.dvbs2hip_filter(self.h, _ptr(X), _ptr(out), X.size // (2 * n_frames), n_frames))
        return out

    def filter_dev(self, X, Y, n_cplx, n_frames):
        self._chk(self.L.dvbs2hip_filter_dev(self.h, _ptr(X), _ptr(Y), n_cplx, n_frames))

    def filter_reset(self):
        self._chk(self.L.dvbs2hip_filter_reset(self.h))

    def filter_split(self, n_cplx):
        """first complex sample of a frame that filter2 (and not filter1) produces"""
        r = self.L.dvbs2hip_filter_split(self.h, n_cplx)
        if r < 0:
            raise ValueError("half a frame has to hold the filter's memory")
        return r

    def filter1(self, X_N1, n_frames=1):
        """Filter<R>::filter1 (Filter_FIR_ccr.cpp:144-218): lower part of every frame, state advanced"""
        X = np.ascontiguousarray(X_N1, dtype=np.float32).ravel()
        if X.size % (2 * n_frames):
            raise ValueError("filter socket size must be a multiple of 2 * n_frames")
        out = np.zeros_like(X)
        self._chk(self.L.dvbs2hip_filter1(self.h, _ptr(X), _ptr(out), X.size // (2 * n_frames), n_frames))
        return out

    def filter2(self, X_N1, Y_N2h, n_frames=1):
        """Filter<R>::filter2 (Filter_FIR_ccr.cpp:220-294): Y_N2h copied, upper part of every frame from X_N1 alone"""
        X = np.ascontiguousarray(X_N1, dtype=np.float32).ravel()
        Yh = np.ascontiguousarray(Y_N2h, dtype=np.float32).ravel()
        if X.size % (2 * n_frames) or Yh.size != X.size:
            raise ValueError("filter2 sockets must have the same size, a multiple of 2 * n_frames")
        out = np.empty_like(X)
        self._chk(self.L.dvbs2hip_filter2(self.h, _ptr(X), _ptr(Yh), _ptr(out), X.size // (2 * n_frames), n_frames))
        return out

    # ------------------------------------------------------------------ a6
    def estimate(self, X_N):
        X, F = self._frames(X_N, 2 * self.N_xfec, np.float32)
        sig, eb, es = (np.empty(F, dtype=np.float32) for _ in range(3))
        self._chk(self.L.dvbs2hip_estimate(self.h, _ptr(X), _ptr(sig), _ptr(eb), _ptr(es), F))
        return sig, eb, es

    # ------------------------------------------------------------------ a7
    def agc(self, X_N, n_frames=1, output_energy=1.0):
        """Multiplier_AGC_cc_naive::imultiply: every frame over its standard deviation (x sqrt(output_energy)); frames of any length"""
        X = np.ascontiguousarray(X_N, dtype=np.float32).ravel()
        if X.size % (2 * n_frames):
            raise ValueError("the socket does not hold %d frames of complex samples" % n_frames)
        Z = np.empty_like(X)
        self._chk(self.L.dvbs2hip_agc_imultiply(self.h, _ptr(X), _ptr(Z), X.size // (2 * n_frames), float(output_energy), n_frames))
        return Z

    def agc_dev(self, X, Z, n_cplx, output_energy, n_frames):
        self._chk(self.L.dvbs2hip_agc_imultiply_dev(self.h, _ptr(X), _ptr(Z), n_cplx, float(output_energy), n_frames))

    def sync_coarse_set_freq(self, estimated_freq):
        self._chk(self.L.dvbs2hip_sync_coarse_set_freq(self.h, float(estimated_freq)))

    def sync_coarse_reset(self):
        self._chk(self.L.dvbs2hip_sync_coarse_reset(self.h))

    def sync_coarse_synchronize(self, X_N1, n_frames=1):
        """the coarse frequency synchronizer's task in the transmission phase: the stream times exp(-j 2 pi estimated_freq n) -> (FRQ[F], PHS[F], Y_N2)"""
        X = np.ascontiguousarray(X_N1, dtype=np.float32).ravel()
        if X.size % (2 * n_frames):
            raise ValueError("the socket does not hold %d frames of complex samples" % n_frames)
        FRQ, PHS, Y = np.empty(n_frames, np.float32), np.empty(n_frames, np.float32), np.empty_like(X)
        self._chk(self.L.dvbs2hip_sync_coarse_synchronize(self.h, _ptr(X), _ptr(FRQ), _ptr(PHS), _ptr(Y), X.size // (2 * n_frames), n_frames))
        return FRQ, PHS, Y

    def pl_descramble(self, Y_N1):
        X, F = self._frames(Y_N1, 2 * self.pl_frame, np.float32)
        out = np.empty_like(X)
        self._chk(self.L.dvbs2hip_pl_descramble(self.h, _ptr(X), _ptr(out), F))
        return out

    def remove_plh(self, Y_N1):
        X, F = self._frames(Y_N1, 2 * self.pl_frame, np.float32)
        out = np.empty((F, 2 * self.N_xfec), dtype=np.float32)
        self._chk(self.L.dvbs2hip_remove_plh(self.h, _ptr(X), _ptr(out), F))
        return out

    # ------------------------------------------------------------------ a8
    def bb_descramble(self, Y_N1):
        X, F = self._frames(Y_N1, self.K_bch, np.int32)
        out = np.empty_like(X)
        self._chk(self.L.dvbs2hip_bb_descramble(self.h, _ptr(X), _ptr(out), F))
        return out

    # ------------------------------------------------------------------ a9
    def check_errors(self, U, V):
        Ua, F = self._frames(U, self.K_bch, np.int32)
        Va, F2 = self._frames(V, self.K_bch, np.int32)
        if F != F2:
            raise ValueError("U and V hold a different number of frames")
        self._chk(self.L.dvbs2hip_monitor_check_errors(self.h, _ptr(Ua), _ptr(Va), F))

    def check_errors_dev(self, U, V, n_frames):
        self._chk(self.L.dvbs2hip_monitor_check_errors_dev(self.h, _ptr(U), _ptr(V), n_frames))

    def monitor_get(self):
        out = np.zeros(3, dtype=np.uint64)
        self._chk(self.L.dvbs2hip_monitor_get(self.h, _ptr(out)))
        return int(out[0]), int(out[1]), int(out[2])     # FRA, BE, FE

    def monitor_reset(self):
        self._chk(self.L.dvbs2hip_monitor_reset(self.h))

    def check_errors2(self, U, V):
        """Monitor_BFER::check_errors2: -> FRA int64[F], BE int32[F], FE int32[F], BER f32[F], FER f32[F] (counters after each frame)"""
        Ua, F = self._frames(U, self.K_bch, np.int32)
        Va, F2 = self._frames(V, self.K_bch, np.int32)
        if F != F2:
            raise ValueError("U and V hold a different number of frames")
        fra = np.empty(F, np.int64); be = np.empty(F, np.int32); fe = np.empty(F, np.int32)
        ber = np.empty(F, np.float32); fer = np.empty(F, np.float32)
        self._chk(self.L.dvbs2hip_monitor_check_errors2(self.h, _ptr(Ua), _ptr(Va), _ptr(fra), _ptr(be), _ptr(fe), _ptr(ber), _ptr(fer), F))
        return fra, be, fe, ber, fer

    def monitor_reduce_init(self, rank, world_size, rendezvous_path=None, timeout_ms=60000):
        self._chk(self.L.dvbs2hip_monitor_reduce_init(self.h, rank, world_size, rendezvous_path.encode() if rendezvous_path else None, timeout_ms))

    def monitor_reduce(self):
        """tools::Monitor_reduction across one-process-per-GPU ranks: RCCL sum of {FRA, BE, FE} (collective)"""
        out = np.zeros(3, dtype=np.uint64)
        self._chk(self.L.dvbs2hip_monitor_reduce(self.h, _ptr(out)))
        return int(out[0]), int(out[1]), int(out[2])

    def monitor_reduce_finalize(self):
        self._chk(self.L.dvbs2hip_monitor_reduce_finalize(self.h))

    # ------------------------------------------------------------------ fused chain
    def rx_bb(self, pl_frames, sigma=None, out=None):
        """out = (info int32[F,K_bch], cwd_ldpc int8[F], cwd_bch int8[F]) reuses the caller's socket buffers"""
        X, F = self._frames(pl_frames, 2 * self.pl_frame, np.float32)
        if out is not None:
            info, c0, c1 = out
            assert info.dtype == np.int32 and info.size == F * self.K_bch and info.flags.c_contiguous and c0.dtype == np.int8 and c1.dtype == np.int8
        else:
            info = np.empty((F, self.K_bch), dtype=np.int32)
            c0, c1 = np.zeros(F, dtype=np.int8), np.zeros(F, dtype=np.int8)
        sg = None
        if sigma is not None:
            sg = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float32).ravel(), (F,)))
        self._chk(self.L.dvbs2hip_rx_bb(self.h, _ptr(X), _ptr(sg), _ptr(info), _ptr(c0), _ptr(c1), F))
        return info, c0, c1

    def rx_bb_dev(self, pl_frames, sigma, info, cwd_ldpc, cwd_bch, n_frames):
        self._chk(self.L.dvbs2hip_rx_bb_dev(self.h, _ptr(pl_frames), _ptr(sigma), _ptr(info), _ptr(cwd_ldpc),
                                            _ptr(cwd_bch), n_frames))

    # ------------------------------------------------------------------ N1: TX mirror + AWGN
    def tx_bb(self, n_frames, info=None, seed=0, sigma=None):
        """-> (info_sent[F,K_bch] int32, pl_frames[F, 2*pl_frame] f32)"""
        F = int(n_frames)
        inf = None
        if info is not None:
            inf, F2 = self._frames(info, self.K_bch, np.int32)
            if F2 != F:
                raise ValueError("info holds %d frames, expected %d" % (F2, F))
        sg = None
        if sigma is not None:
            sg = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float32).ravel(), (F,)))
        sent = np.empty((F, self.K_bch), dtype=np.int32)
        pl = np.empty((F, 2 * self.pl_frame), dtype=np.float32)
        self._chk(self.L.dvbs2hip_tx_bb(self.h, _ptr(inf), int(seed), _ptr(sg), _ptr(sent), _ptr(pl), F))
        return sent, pl

    def tx_bb_dev(self, info_in, seed, sigma, info_out, pl_frames, n_frames):
        self._chk(self.L.dvbs2hip_tx_bb_dev(self.h, _ptr(info_in), int(seed), _ptr(sigma), _ptr(info_out), _ptr(pl_frames), n_frames))

    # ------------------------------------------------------------------ N2: shaping filter, noise, perfect timing
    def shape_filter(self, X_N1, n_frames=1, osf=2):
        X = np.ascontiguousarray(X_N1, dtype=np.float32).ravel()
        out = np.empty(X.size * osf, dtype=np.float32)
        self._chk(self.L.dvbs2hip_shape_filter(self.h, _ptr(X), _ptr(out), X.size // (2 * n_frames), n_frames))
        return out

    def shape_filter_dev(self, X, Y, n_cplx, n_frames):
        self._chk(self.L.dvbs2hip_shape_filter_dev(self.h, _ptr(X), _ptr(Y), n_cplx, n_frames))

    def add_noise(self, sigma, X_N, seed=0, n_frames=1):
        X = np.ascontiguousarray(X_N, dtype=np.float32).ravel()
        sg = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float32).ravel(), (n_frames,)))
        out = np.empty_like(X)
        self._chk(self.L.dvbs2hip_add_noise(self.h, _ptr(sg), _ptr(X), _ptr(out), int(seed), X.size // n_frames, n_frames))
        return out

    def add_noise_dev(self, sigma, X, Y, seed, n_elmts, n_frames):
        self._chk(self.L.dvbs2hip_add_noise_dev(self.h, _ptr(sigma), _ptr(X), _ptr(Y), int(seed), n_elmts, n_frames))

    def graph_capture(self, fn):
        """records the _dev calls `fn` makes on this handle as one hipGraph -> graph id (dvbs2hip_graph_begin / _end); run the sequence once before recording it"""
        self._chk(self.L.dvbs2hip_graph_begin(self.h))
        try:
            fn()
        finally:
            g = C.c_int32(-1)
            rc = self.L.dvbs2hip_graph_end(self.h, C.byref(g))
        self._chk(rc)
        return g.value

    def graph_launch(self, graph):
        self._chk(self.L.dvbs2hip_graph_launch(self.h, int(graph)))

    def graph_destroy(self, graph):
        self._chk(self.L.dvbs2hip_graph_destroy(self.h, int(graph)))

    def extract_dev(self, X, Y, n_cplx_out, osf, offset, n_frames):
        self._chk(self.L.dvbs2hip_extract_dev(self.h, _ptr(X), _ptr(Y), n_cplx_out, osf, int(offset), n_frames))

    # ------------------------------------------------------------------ measurement
    # ------------------------------------------------------------------ N4: frame synchronizer
    def sync_frame_set_params(self, alpha=0.9, trigger=30.0, vec_width=8):
        self._chk(self.L.dvbs2hip_sync_frame_set_params(self.h, float(alpha), float(trigger), int(vec_width)))

    def sync_frame_reset(self):
        self._chk(self.L.dvbs2hip_sync_frame_reset(self.h))

    def sync_frame_synchronize1(self, X_N1):
        X, F = self._frames(X_N1, 2 * self.pl_frame, np.float32)
        cs, cp = np.empty_like(X), np.empty_like(X)
        self._chk(self.L.dvbs2hip_sync_frame_synchronize1(self.h, _ptr(X), _ptr(cs), _ptr(cp), F))
        return cs, cp

    def sync_frame_synchronize2(self, X_N1, cor_sof, cor_plsc, with_flags=False):
        X, F = self._frames(X_N1, 2 * self.pl_frame, np.float32)
        cs, _ = self._frames(cor_sof, 2 * self.pl_frame, np.float32)
        cp, _ = self._frames(cor_plsc, 2 * self.pl_frame, np.float32)
        DEL, FLG, TRI, Y = np.empty(F, np.int32), np.empty(F, np.int32), np.empty(F, np.float32), np.empty_like(X)
        self._chk(self.L.dvbs2hip_sync_frame_synchronize2(self.h, _ptr(X), _ptr(cs), _ptr(cp), _ptr(DEL), _ptr(FLG), _ptr(TRI), _ptr(Y), F))
        return (DEL, FLG, TRI, Y) if with_flags else (DEL, Y)

    def sync_frame_synchronize(self, X_N1, with_flags=False):
        X, F = self._frames(X_N1, 2 * self.pl_frame, np.float32)
        DEL, FLG, TRI, Y = np.empty(F, np.int32), np.empty(F, np.int32), np.empty(F, np.float32), np.empty_like(X)
        self._chk(self.L.dvbs2hip_sync_frame_synchronize(self.h, _ptr(X), _ptr(DEL), _ptr(FLG), _ptr(TRI), _ptr(Y), F))
        return (DEL, FLG, TRI, Y) if with_flags else (DEL, Y)

    def sync_frame_synchronize_dev(self, X, DEL, FLG, TRI, Y, n_frames):
        self._chk(self.L.dvbs2hip_sync_frame_synchronize_dev(self.h, X, DEL, FLG, TRI, Y, n_frames))

    def sync_frame_locate_dev(self, X, DEL, FLG, TRI, SRC, n_frames):
        """chained device form: SRC (device table of n_frames device pointers) <- where each aligned frame starts (inside X or the handle's scratch)"""
        self._chk(self.L.dvbs2hip_sync_frame_locate_dev(self.h, X, DEL, FLG, TRI, SRC, n_frames))

    def rx_bb_located_dev(self, SRC, sigma, info, cwd_ldpc, cwd_bch, n_frames):
        self._chk(self.L.dvbs2hip_rx_bb_located_dev(self.h, SRC, _ptr(sigma), _ptr(info), _ptr(cwd_ldpc), _ptr(cwd_bch), n_frames))

    def sync_frame_metric(self):
        m, fl = C.c_float(), C.c_int32()
        self._chk(self.L.dvbs2hip_sync_frame_get_metric(self.h, C.byref(m), C.byref(fl)))
        return m.value, bool(fl.value)

    # ------------------------------------------------------------------ N4: fine frequency / phase synchronizers
    def _sff(self, fn, X_N1):
        X, F = self._frames(X_N1, 2 * self.pl_frame, np.float32)
        FRQ, PHS, Y = np.empty(F, np.float32), np.empty(F, np.float32), np.empty_like(X)
        self._chk(fn(self.h, _ptr(X), _ptr(FRQ), _ptr(PHS), _ptr(Y), F))
        return FRQ, PHS, Y

    def sync_lr_synchronize(self, X_N1):
        return self._sff(self.L.dvbs2hip_sync_lr_synchronize, X_N1)

    def sync_lr_set_alpha(self, alpha):
        self._chk(self.L.dvbs2hip_sync_lr_set_alpha(self.h, float(alpha)))

    def sync_lr_reset(self):
        self._chk(self.L.dvbs2hip_sync_lr_reset(self.h))

    def sync_lr_timeouts(self) -> int:
        """launches of the fused L&R kernel whose rotation had to be repeated (a waiting workgroup gave up): 0 unless the dispatcher reorders workgroups"""
        n = C.c_int32()
        self._chk(self.L.dvbs2hip_sync_lr_timeouts(self.h, C.byref(n)))
        return n.value

    def sync_freq_phase_synchronize(self, X_N1):
        return self._sff(self.L.dvbs2hip_sync_freq_phase_synchronize, X_N1)

    def set_filter_kernel(self, kernel):
        """B.FIR_AUTO (default: matrix cores for <= 81 taps), B.FIR_VALU or B.FIR_MFMA"""
        self._chk(self.L.dvbs2hip_set_filter_kernel(self.h, int(kernel)))

    def set_ldpc_schedule(self, schedule):
        """B.SCHED_QC (default) or B.SCHED_NATURAL (the reference's row order, one lane per frame)"""
        self._chk(self.L.dvbs2hip_set_ldpc_schedule(self.h, int(schedule)))

    def ldpc_kernel_name(self) -> str:
        return self.L.dvbs2hip_ldpc_kernel_name(self.h).decode()

    def timing_enable(self, on=True):
        self._chk(self.L.dvbs2hip_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        self._chk(self.L.dvbs2hip_timing_reset(self.h))

    def device_copy_GBps(self, nbytes=1 << 30, reps=5) -> float:
        """read + written GB/s of the library's own streaming copy kernel on this GPU (a measurement aid: dvbs2hip_device_copy_bandwidth)"""
        g = C.c_double()
        self._chk(self.L.dvbs2hip_device_copy_bandwidth(self.h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    def timing_stats(self):
        """-> {group: (device ms, launches)} of the per-kernel timers, the groups of `--sim-stats` (include/dvbs2hip.h DVBS2HIP_K_*)"""
        names = ("LDPC decoder", "BCH decoder", "demodulator", "filters", "front end (descramble + estimate + demodulate)", "other (TX mirror, synchronizers, monitor)")
        return {n: self.timing_get(k) for k, n in enumerate(names)}

    def timing_get(self, k):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self.L.dvbs2hip_timing_get(self.h, k, C.byref(ms), C.byref(n)))
        return ms.value, n.value
