"""ctypes front-end of the CPU ORACLE (oracle/libdvbs2_oracle.so).  TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
from the dvbs2_amd package (tests/test_no_oracle_in_product.py enforces that).
Parity status: see oracle/dvbs2_oracle.h ("parity unpinned" for a1-a4).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdvbs2_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dvbs2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_NATIVE_SO = os.path.join(_HERE, "_native", "libdvbs2_oracle_native.so")


def build_native() -> str:
    """The CPU baseline's -march=native build (bench.py's cpu_baseline leg): always rebuilt by make on the host that runs it when the
    source is newer; raises when the host has no compiler."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "native"])
    return _NATIVE_SO


def native_isa() -> str:
    """widest vector registers the native build's inter-frame loop uses: 'zmm' / 'ymm' / 'xmm' / '?' (objdump of decode_inter_block)"""
    import shutil
    if not shutil.which("objdump") or not os.path.exists(_NATIVE_SO):
        return "?"
    dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", _NATIVE_SO], capture_output=True, text=True).stdout
    body, on = [], False
    for line in dis.splitlines():
        if line.endswith("<decode_inter_block>:"):
            on = True
        elif on and line.strip() == "":
            break
        elif on:
            body.append(line)
    for reg in ("zmm", "ymm", "xmm"):
        if any(reg in l and ("vsubps" in l or "vaddps" in l or "vcmp" in l) for l in body):
            return reg
    return "?"


_lib = None
_libs = {}


def _bind(path):
    if path not in _libs:
        L = C.CDLL(path)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.orc_ldpc_create.restype = vp
        L.orc_ldpc_create.argtypes = [ci, ci, ci, vp, vp]
        L.orc_ldpc_destroy.argtypes = [vp]
        L.orc_ldpc_decode_batch.restype = C.c_double
        L.orc_ldpc_decode_batch.argtypes = [vp, vp, ci, ci, ci, cf, vp, ci]
        L.orc_ldpc_decode_batch_inter.restype = C.c_double
        L.orc_ldpc_decode_batch_inter.argtypes = [vp, vp, ci, ci, cf, vp, ci]
        L.orc_ldpc_inter_width.restype = ci
        L.orc_ldpc_decode_batch_pinned.restype = C.c_double
        L.orc_ldpc_decode_batch_pinned.argtypes = [vp, vp, ci, ci, ci, cf, vp, ci, vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_stream_triad_GBps.restype = C.c_double
        L.orc_stream_triad_GBps.argtypes = [ci, C.c_size_t, ci, vp, ci]
        _libs[path] = L
    return _libs[path]


class NativeLdpc:
    """The batch decoders of the -march=native build (CPU baseline only): same source, same results as Ldpc.decode_batch*_timed."""
    def __init__(self, N, K, row_ptr, addr):
        self.L = _bind(build_native())
        self.N, self.K = N, K
        self.inter_width = self.L.orc_ldpc_inter_width()
        self._rp, self._ad = _i32(row_ptr), _i32(addr)
        self.h = self.L.orc_ldpc_create(N, K, len(self._rp) - 1, _p(self._rp), _p(self._ad))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_ldpc_destroy(self.h)
            self.h = None

    def decode_batch_timed(self, llr, n_ite=10, alpha=1.0, sched=0, threads=1):
        llr = _f32(llr).reshape(-1, self.N)
        bits = np.empty((llr.shape[0], self.K), dtype=np.int32)
        return bits, self.L.orc_ldpc_decode_batch(self.h, _p(llr), llr.shape[0], sched, n_ite, alpha, _p(bits), threads)

    def decode_batch_inter_timed(self, llr, n_ite=10, alpha=1.0, threads=1):
        llr = _f32(llr).reshape(-1, self.N)
        bits = np.empty((llr.shape[0], self.K), dtype=np.int32)
        return bits, self.L.orc_ldpc_decode_batch_inter(self.h, _p(llr), llr.shape[0], n_ite, alpha, _p(bits), threads)

    def decode_batch_pinned(self, llr, flavour, n_ite=10, alpha=1.0, threads=1, cpus=None, want_bits=False):
        """bench.py's CPU baseline: pinned threads, first-touched private buffers, static deal -> (seconds, slowest thread's seconds, fastest thread's seconds[, bits])"""
        llr = _f32(llr).reshape(-1, self.N)
        bits = np.empty((llr.shape[0], self.K), dtype=np.int32) if want_bits else None
        cp = _i32(cpus) if cpus is not None and len(cpus) else None
        hi, lo = C.c_double(), C.c_double()
        sec = self.L.orc_ldpc_decode_batch_pinned(self.h, _p(llr), llr.shape[0], int(flavour), n_ite, alpha, _p(bits) if want_bits else None, threads,
                                                  _p(cp) if cp is not None else None, len(cp) if cp is not None else 0, C.byref(hi), C.byref(lo))
        return (sec, hi.value, lo.value, bits) if want_bits else (sec, hi.value, lo.value)

    def stream_triad_GBps(self, threads, floats_per_thread=1 << 24, reps=3, cpus=None):
        cp = _i32(cpus) if cpus is not None and len(cpus) else None
        return self.L.orc_stream_triad_GBps(threads, floats_per_thread, reps, _p(cp) if cp is not None else None, len(cp) if cp is not None else 0)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.orc_ldpc_create.restype = vp
        L.orc_ldpc_create.argtypes = [ci, ci, ci, vp, vp]
        L.orc_ldpc_destroy.argtypes = [vp]
        L.orc_ldpc_n_edges.argtypes = [vp]
        L.orc_ldpc_q.argtypes = [vp]
        L.orc_ldpc_csr.argtypes = [vp, vp, vp]
        L.orc_ldpc_encode.argtypes = [vp, vp, vp]
        L.orc_ldpc_syndrome_weight.argtypes = [vp, vp]
        L.orc_ldpc_decode.argtypes = [vp, vp, ci, ci, ci, cf, ci, vp, vp, vp]
        L.orc_det_tanh_half.restype = cf
        L.orc_det_tanh_half.argtypes = [cf]
        L.orc_det_log1p.restype = cf
        L.orc_det_log1p.argtypes = [cf]
        L.orc_chk_update.argtypes = [ci, cf, vp, ci, vp]
        L.orc_ldpc_decode_batch.restype = C.c_double
        L.orc_ldpc_decode_batch.argtypes = [vp, vp, ci, ci, ci, cf, vp, ci]
        L.orc_ldpc_decode_batch_inter.restype = C.c_double
        L.orc_ldpc_decode_batch_inter.argtypes = [vp, vp, ci, ci, cf, vp, ci]
        L.orc_bch_create.restype = vp
        L.orc_bch_create.argtypes = [ci, vp, ci, ci, ci]
        L.orc_bch_destroy.argtypes = [vp]
        L.orc_bch_gen_degree.argtypes = [vp]
        L.orc_bch_gen.restype = vp
        L.orc_bch_gen.argtypes = [vp]
        L.orc_bch_encode.argtypes = [vp, vp, vp]
        L.orc_bch_decode.argtypes = [vp, vp, vp, vp]
        L.orc_cstl_normalise.argtypes = [vp, ci, vp]
        L.orc_modulate.argtypes = [vp, ci, vp, ci, vp]
        L.orc_demodulate.argtypes = [vp, ci, cf, vp, ci, vp]
        L.orc_itl_lut.argtypes = [ci, ci, ci, vp]
        L.orc_pl_rand_seq.argtypes = [ci, vp]
        L.orc_pl_scramble.argtypes = [vp, vp, ci, ci, ci]
        L.orc_bb_scramble.argtypes = [vp, vp, ci]
        L.orc_plheader.argtypes = [vp, vp]
        L.orc_framer_generate.argtypes = [vp, ci, vp, vp]
        L.orc_framer_remove_plh.argtypes = [vp, ci, vp]
        L.orc_pl_frame_size.argtypes = [ci]
        L.orc_estimate.argtypes = [vp, ci, cf, ci, vp]
        L.orc_agc.argtypes = [vp, ci, cf, vp]
        L.orc_nco.argtypes = [vp, ci, cf, vp, vp]
        L.orc_rrc_taps.argtypes = [cf, ci, ci, vp]
        L.orc_fir.argtypes = [vp, ci, vp, vp, vp, ci]
        L.orc_upfir.argtypes = [vp, ci, ci, vp, vp, vp, ci]
        L.orc_sff_pilots.argtypes = [ci, vp, ci]
        L.orc_lr_synchronize.restype = cf
        L.orc_lr_synchronize.argtypes = [ci, cf, vp, vp, vp]
        L.orc_fp_synchronize.argtypes = [ci, vp, vp, vp]
        L.orc_vdelay_create.restype = vp
        L.orc_vdelay_create.argtypes = [ci, ci, ci]
        L.orc_vdelay_destroy.argtypes = [vp]
        L.orc_vdelay_set_delay.argtypes = [vp, ci]
        L.orc_vdelay_reset.argtypes = [vp]
        L.orc_vdelay_filter.argtypes = [vp, vp, vp]
        L.orc_sfm_create.restype = vp
        L.orc_sfm_create.argtypes = [ci, cf, cf, ci]
        L.orc_sfm_destroy.argtypes = [vp]
        L.orc_sfm_reset.argtypes = [vp]
        L.orc_sfm_synchronize1.argtypes = [vp, vp, vp, vp]
        L.orc_sfm_synchronize2.argtypes = [vp, vp, vp, vp, vp]
        L.orc_sfm_synchronize.argtypes = [vp, vp, vp]
        L.orc_sfm_metric.restype = cf
        L.orc_sfm_metric.argtypes = [vp]
        L.orc_sfm_packet_flag.argtypes = [vp]
        L.orc_sfm_taps.argtypes = [vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


NMS, SPA, SPA_TANH, SPA_CLIP = 0, 1, 2, 3      # SPA: exact boxplus; SPA_TANH: the saturating tanh-product form of AFF3CT's Update_rule_SPA (dvbs2_oracle.c)
NATURAL, QC, QC_SEQ, QC_FIX = 0, 1, 2, 3


def det_tanh_half(a):
    """tanh(a / 2) as the SPA_TANH rule evaluates it (correctly rounded operations only)"""
    return float(lib().orc_det_tanh_half(float(a)))


def det_log1p(w):
    return float(lib().orc_det_log1p(float(w)))


def chk_update(v2c, implem=SPA_TANH, alpha=1.0):
    """one check node: v->c messages in, c->v messages out"""
    v2c = _f32(v2c)
    out = np.empty_like(v2c)
    lib().orc_chk_update(implem, alpha, _p(v2c), v2c.size, _p(out))
    return out


class Ldpc:
    def __init__(self, N, K, row_ptr, addr):
        self.N, self.K = N, K
        self._rp, self._ad = _i32(row_ptr), _i32(addr)
        self.h = lib().orc_ldpc_create(N, K, len(self._rp) - 1, _p(self._rp), _p(self._ad))
        if not self.h:
            raise ValueError("bad LDPC table")
        self.q = lib().orc_ldpc_q(self.h)
        self.E = lib().orc_ldpc_n_edges(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ldpc_destroy(self.h)
            self.h = None

    def csr(self):
        a, b = C.c_void_p(), C.c_void_p()
        lib().orc_ldpc_csr(self.h, C.byref(a), C.byref(b))
        M = self.N - self.K
        ptr = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_int)), (M + 1,)).copy()
        var = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_int)), (self.E,)).copy()
        return ptr, var

    def encode(self, info):
        info = _i32(info).reshape(-1, self.K)
        cw = np.empty((info.shape[0], self.N), dtype=np.int32)
        for f in range(info.shape[0]):
            lib().orc_ldpc_encode(self.h, _p(info[f]), _p(cw[f]))
        return cw

    def syndrome_weight(self, cw):
        cw = _i32(cw)
        return lib().orc_ldpc_syndrome_weight(self.h, _p(cw))

    def decode(self, llr, n_ite=10, alpha=1.0, implem=NMS, sched=QC, early_stop=False):
        """-> bits[F,K] int32, post[F,N] f32, cwd[F] int8, n_ite_done[F]"""
        llr = _f32(llr).reshape(-1, self.N)
        F = llr.shape[0]
        bits = np.empty((F, self.K), dtype=np.int32)
        post = np.empty((F, self.N), dtype=np.float32)
        cwd = np.zeros(F, dtype=np.int8)
        ites = np.zeros(F, dtype=np.int32)
        for f in range(F):
            ites[f] = lib().orc_ldpc_decode(self.h, _p(llr[f]), implem, sched, n_ite, alpha,
                                            int(early_stop), _p(bits[f]), _p(post[f]),
                                            C.c_void_p(cwd.ctypes.data + f))
        return bits, post, cwd, ites

    def decode_batch_timed(self, llr, n_ite=10, alpha=1.0, sched=NATURAL, threads=1):
        llr = _f32(llr).reshape(-1, self.N)
        bits = np.empty((llr.shape[0], self.K), dtype=np.int32)
        sec = lib().orc_ldpc_decode_batch(self.h, _p(llr), llr.shape[0], sched, n_ite, alpha,
                                          _p(bits), threads)
        return bits, sec


    def decode_batch_inter_timed(self, llr, n_ite=10, alpha=1.0, threads=1):
        """inter-frame SIMD CPU flavour (8 or 16 frames per vector), natural order NMS"""
        llr = _f32(llr).reshape(-1, self.N)
        bits = np.empty((llr.shape[0], self.K), dtype=np.int32)
        sec = lib().orc_ldpc_decode_batch_inter(self.h, _p(llr), llr.shape[0], n_ite, alpha, _p(bits), threads)
        return bits, sec


class Bch:
    def __init__(self, m, prim, t, N, K):
        self.m, self.t, self.N, self.K = m, t, N, K
        self._prim = _i32(prim)
        self.h = lib().orc_bch_create(m, _p(self._prim), t, N, K)
        self.gdeg = lib().orc_bch_gen_degree(self.h)
        if self.gdeg != N - K:
            raise ValueError("generator degree %d != N-K %d" % (self.gdeg, N - K))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_bch_destroy(self.h)
            self.h = None

    def gen(self):
        p = lib().orc_bch_gen(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int)), (self.gdeg + 1,)).copy()

    def encode(self, info):
        info = _i32(info).reshape(-1, self.K)
        cw = np.empty((info.shape[0], self.N), dtype=np.int32)
        for f in range(info.shape[0]):
            lib().orc_bch_encode(self.h, _p(info[f]), _p(cw[f]))
        return cw

    def decode(self, cw):
        cw = _i32(cw).reshape(-1, self.N)
        out = np.empty((cw.shape[0], self.K), dtype=np.int32)
        cwd = np.zeros(cw.shape[0], dtype=np.int8)
        for f in range(cw.shape[0]):
            lib().orc_bch_decode(self.h, _p(cw[f]), _p(out[f]), C.c_void_p(cwd.ctypes.data + f))
        return out, cwd


def cstl_normalise(pts):
    pts = _f32(pts).reshape(-1, 2)
    out = np.empty_like(pts)
    lib().orc_cstl_normalise(_p(pts), pts.shape[0], _p(out))
    return out


def modulate(cstl, bps, bits):
    cstl, bits = _f32(cstl), _i32(bits).ravel()
    sym = np.empty(2 * (bits.size // bps), dtype=np.float32)
    lib().orc_modulate(_p(cstl), bps, _p(bits), bits.size, _p(sym))
    return sym


def demodulate(cstl, bps, sigma, sym):
    cstl, sym = _f32(cstl), _f32(sym).ravel()
    n_sym = sym.size // 2
    llr = np.empty(n_sym * bps, dtype=np.float32)
    lib().orc_demodulate(_p(cstl), bps, float(sigma), _p(sym), n_sym, _p(llr))
    return llr


def itl_lut(N, n_cols, order):
    lut = np.empty(N, dtype=np.uint32)
    lib().orc_itl_lut(N, n_cols, order, _p(lut))
    return lut


def pl_rand_seq(n=0):
    seq = np.empty(66420, dtype=np.uint8)
    lib().orc_pl_rand_seq(n, _p(seq))
    return seq


def pl_scramble(x, start_ix=90, scramble=True):
    x = _f32(x).ravel()
    y = np.empty_like(x)
    lib().orc_pl_scramble(_p(x), _p(y), x.size // 2, start_ix, int(scramble))
    return y


def bb_scramble(bits):
    bits = _i32(bits).ravel()
    out = np.empty_like(bits)
    lib().orc_bb_scramble(_p(bits), _p(out), bits.size)
    return out


def plheader(mod_cod7):
    mc = _i32(mod_cod7)
    plh = np.empty(180, dtype=np.float32)
    lib().orc_plheader(_p(mc), _p(plh))
    return plh


def pl_frame_size(n_xfec_sym):
    return lib().orc_pl_frame_size(n_xfec_sym)


def framer_generate(xfec, plh):
    xfec, plh = _f32(xfec).ravel(), _f32(plh)
    n = xfec.size // 2
    out = np.empty(2 * pl_frame_size(n), dtype=np.float32)
    lib().orc_framer_generate(_p(xfec), n, _p(plh), _p(out))
    return out


def framer_remove_plh(plf, n_xfec_sym):
    plf = _f32(plf).ravel()
    out = np.empty(2 * n_xfec_sym, dtype=np.float32)
    lib().orc_framer_remove_plh(_p(plf), n_xfec_sym, _p(out))
    return out


def estimate(xfec, code_rate, bps):
    xfec = _f32(xfec).ravel()
    out = np.empty(3, dtype=np.float32)
    lib().orc_estimate(_p(xfec), xfec.size // 2, float(code_rate), bps, _p(out))
    return out   # sigma, ebn0, esn0


def agc(x, output_energy=1.0):
    """Multiplier_AGC_cc_naive::_imultiply on ONE frame of interleaved complex samples"""
    x = _f32(x).ravel()
    z = np.empty_like(x)
    lib().orc_agc(_p(x), x.size // 2, float(output_energy), _p(z))
    return z


def nco(x, nu, n0=0.0):
    """Multiplier_sine_ccc_naive::imultiply on a stretch of the stream that starts at position n0 -> (z, position behind it)"""
    x = _f32(x).ravel()
    z = np.empty_like(x)
    n = np.array([n0], dtype=np.float32)
    lib().orc_nco(_p(x), x.size // 2, float(nu), _p(n), _p(z))
    return z, float(n[0])


def rrc_taps(rolloff=0.2, osf=2, grp_delay=20):
    t = np.empty(2 * grp_delay * osf + 1, dtype=np.float32)
    lib().orc_rrc_taps(float(rolloff), osf, grp_delay, _p(t))
    return t


def fir(taps, hist, x):
    """hist: float32[2*(T-1)] updated in place.  x: interleaved complex."""
    taps, x = _f32(taps), _f32(x).ravel()
    y = np.empty_like(x)
    assert hist.dtype == np.float32 and hist.size == 2 * (taps.size - 1)
    lib().orc_fir(_p(taps), taps.size, _p(hist), _p(x), _p(y), x.size // 2)
    return y


def fir_split(n_floats, n_taps, simd_width=8):
    """Where Filter_FIR_ccr::_filter1 stops and ::_filter2 starts, in FLOATS of one frame of n_floats (Filter_FIR_ccr.cpp:
    rest = N - P*M with M = mipp::N<R>() and P = (N - 2(T-1)) / M, .cpp:17-18,152; filter1 runs i = rest .. N/2 step M (:183),
    filter2 from up_half = N/2 + ((N/2 - rest) % M) (:243-245)).  -> (end of what filter1 wrote, up_half): the first exceeds or
    equals the second, so together they cover the frame."""
    M = simd_width
    P = (n_floats - 2 * (n_taps - 1)) // M
    rest = n_floats - P * M
    last = rest + ((n_floats // 2 - 1 - rest) // M) * M          # last i < N/2 of filter1's vector loop
    up_half = n_floats // 2 + ((n_floats // 2 - rest) % M)
    return last + M, up_half


def fir1(taps, hist, x, n_frames=1, simd_width=8):
    """Filter<R>::filter1 over n_frames consecutive frames (Filter.hxx:149-160 loops _filter1 over the frames;
    Filter_FIR_ccr.cpp:144-218): every frame's outputs below the split, the rest of Y_N2 untouched (zeros here);
    the state (hist) advances by whole frames exactly as in fir()."""
    taps, x = _f32(taps), _f32(x).ravel()
    n = x.size // n_frames
    y = np.zeros_like(x)
    full = fir(taps, hist, x)
    end1, _ = fir_split(n, taps.size, simd_width)
    for f in range(n_frames):
        y[f * n: f * n + end1] = full[f * n: f * n + end1]
    return y


def fir2(taps, x, yh, n_frames=1, simd_width=8):
    """Filter<R>::filter2 (Filter_FIR_ccr.cpp:220-294): Y_N2 = Y_N2h, then the outputs from up_half on, computed from X_N1
    alone (they read nothing before the frame: up_half >= 2 (T-1)); no state."""
    taps, x, yh = _f32(taps), _f32(x).ravel(), _f32(yh).ravel()
    n = x.size // n_frames
    y = yh.copy()
    _, up = fir_split(n, taps.size, simd_width)
    assert up >= 2 * (taps.size - 1)
    for f in range(n_frames):
        full = fir(taps, np.zeros(2 * (taps.size - 1), np.float32), x[f * n:(f + 1) * n])
        y[f * n + up:(f + 1) * n] = full[up:]
    return y


def upfir(taps, osf, hist, x):
    taps, x = _f32(taps), _f32(x).ravel()
    assert hist.dtype == np.float32 and hist.size == 2 * (taps.size - 1)      # orc_upfir filters the zero-stuffed stream: the history is in OUTPUT samples
    y = np.empty(x.size * osf, dtype=np.float32)
    lib().orc_upfir(_p(taps), taps.size, osf, _p(hist), _p(x), _p(y), x.size // 2)
    return y


def sff_pilots(n_cplx):
    ps = np.zeros(64, dtype=np.int32)
    n = lib().orc_sff_pilots(n_cplx, _p(ps), 64)
    return ps[:n].copy()


class SyncLR:
    """Synchronizer_Luise_Reggiannini_DVBS2_aib (row N4): fine frequency from the pilots, damped across frames."""

    def __init__(self, n_cplx, alpha=0.999):
        self.n, self.alpha = n_cplx, float(alpha)
        self.R_l = np.zeros(2, dtype=np.float32)

    def reset(self):
        self.R_l[:] = 0

    def synchronize(self, X):
        X = _f32(X).ravel()
        assert X.size == 2 * self.n
        Y = np.empty_like(X)
        frq = lib().orc_lr_synchronize(self.n, self.alpha, _p(self.R_l), _p(X), _p(Y))
        return float(frq), 0.0, Y


def sync_freq_phase(X):
    """Synchronizer_freq_phase_DVBS2_aib (row N4), stateless -> FRQ, PHS, Y"""
    X = _f32(X).ravel()
    Y, o = np.empty_like(X), np.empty(2, dtype=np.float32)
    lib().orc_fp_synchronize(X.size // 2, _p(X), _p(Y), _p(o))
    return float(o[0]), float(o[1]), Y


class VariableDelay:
    """Variable_delay_cc_naive, block form (row N4).  The output buffer is kept between calls, as the
    reference's socket buffer is."""

    def __init__(self, N, delay, max_delay):
        self.N = N
        self.h = lib().orc_vdelay_create(N, delay, max_delay)
        self.Y = np.zeros(N, dtype=np.float32)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_vdelay_destroy(self.h)
            self.h = None

    def set_delay(self, d):
        lib().orc_vdelay_set_delay(self.h, int(d))

    def filter(self, X):
        X = _f32(X).ravel()
        assert X.size == self.N
        lib().orc_vdelay_filter(self.h, _p(X), _p(self.Y))
        return self.Y.copy()


class SyncFrame:
    """Synchronizer_frame_DVBS2_fast (row N4), one PL frame per call; the output frame is kept between
    calls (the variable delay re-reads it while the delay grows)."""

    def __init__(self, n_cplx, alpha=0.9, trigger=30.0, vec_width=8):
        self.n = n_cplx
        self.h = lib().orc_sfm_create(n_cplx, float(alpha), float(trigger), vec_width)
        self.Y = np.zeros(2 * n_cplx, dtype=np.float32)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sfm_destroy(self.h)
            self.h = None

    def reset(self):
        lib().orc_sfm_reset(self.h)

    def synchronize1(self, X):
        X = _f32(X).ravel()
        assert X.size == 2 * self.n
        cs, cp = np.empty_like(X), np.empty_like(X)
        lib().orc_sfm_synchronize1(self.h, _p(X), _p(cs), _p(cp))
        return cs, cp

    def synchronize2(self, X, cor_sof, cor_plsc):
        X, cs, cp = _f32(X).ravel(), _f32(cor_sof).ravel(), _f32(cor_plsc).ravel()
        d = lib().orc_sfm_synchronize2(self.h, _p(X), _p(cs), _p(cp), _p(self.Y))
        return int(d), self.Y.copy()

    def synchronize(self, X):
        X = _f32(X).ravel()
        assert X.size == 2 * self.n
        d = lib().orc_sfm_synchronize(self.h, _p(X), _p(self.Y))
        return int(d), self.Y.copy()

    @property
    def metric(self):
        return float(lib().orc_sfm_metric(self.h))

    @property
    def packet_flag(self):
        return bool(lib().orc_sfm_packet_flag(self.h))


def sync_frame_taps():
    a, b = C.c_void_p(), C.c_void_p()
    na, nb = C.c_int(), C.c_int()
    lib().orc_sfm_taps(C.byref(a), C.byref(na), C.byref(b), C.byref(nb))
    sof = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_float)), (na.value,)).copy()
    plsc = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_float)), (nb.value,)).copy()
    return sof, plsc


# ---------------------------------------------------------------- convenience: full TX / RX chains
class Chain:
    """Oracle TX and RX baseband chains with the socket graph of
    /root/reference src/mains/TX_RX_BB/main.cpp:75-94."""

    def __init__(self, mc):
        from dvbs2_amd import params as P   # data tables only (no product code paths)
        self.mc = mc
        rp, ad = P.load_ldpc_table(mc.ldpc_table)
        self.ldpc = Ldpc(mc.N_ldpc, mc.K_ldpc, rp, ad)
        self.bch = Bch(mc.bch_m, mc.bch_prim, mc.bch_t, mc.N_bch, mc.K_bch)
        self.cstl = cstl_normalise(P.load_constellation(mc.cstl_file))
        self.lut = itl_lut(mc.N_ldpc, mc.itl_cols, mc.itl_order)
        self.plh = plheader(mc.pls)

    def tx(self, info_bits):
        """info_bits int32[K_bch] -> (pl_frame f32[2*pl_frame] scrambled, ldpc codeword)"""
        mc = self.mc
        scr = bb_scramble(info_bits)
        bch = self.bch.encode(scr)[0]
        cw = self.ldpc.encode(bch)[0]
        itl = cw[self.lut]
        sym = modulate(self.cstl, mc.bps, itl)
        plf = framer_generate(sym, self.plh)
        return pl_scramble(plf, 90, True), cw

    def rx(self, pl_frame, sigma=None, n_ite=10, alpha=1.0, implem=NMS, sched=QC, early_stop=False):
        mc = self.mc
        d = pl_scramble(pl_frame, 90, False)
        xfec = framer_remove_plh(d, mc.N_xfec)
        est = estimate(xfec, mc.code_rate, mc.bps)
        s = float(est[0]) if sigma is None else float(sigma)
        llr_i = demodulate(self.cstl, mc.bps, s, xfec)
        llr = np.empty_like(llr_i)
        llr[self.lut] = llr_i
        bits, post, cwd, ites = self.ldpc.decode(llr, n_ite, alpha, implem, sched, early_stop)
        out, cwd2 = self.bch.decode(bits[0])
        info = bb_scramble(out[0])
        return dict(info=info, llr=llr, post=post[0], ldpc_bits=bits[0], ldpc_cwd=cwd[0],
                    bch_cwd=cwd2[0], sigma=s, est=est, ites=ites[0], xfec=xfec)
