"""ctypes binding of libdvbs2hip.so -- exactly the entry points include/dvbs2hip.h declares.

This is the stub a Python host adds to bind the C ABI (INTEGRATION.md shows the C++/StreamPU
one).  There is no CPU fallback: if the library is missing or no HIP device is present, the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_lib = None


class Dvbs2HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dvbs2hip error %d: %s" % (code, msg))
        self.code = code


class Cfg(C.Structure):
    _fields_ = [
        ("N_ldpc", C.c_int32), ("K_ldpc", C.c_int32), ("K_bch", C.c_int32),
        ("ldpc_n_rows", C.c_int32), ("ldpc_row_ptr", C.c_void_p), ("ldpc_addr", C.c_void_p),
        ("ldpc_n_ite", C.c_int32), ("ldpc_implem", C.c_int32), ("ldpc_alpha", C.c_float),
        ("ldpc_early_stop", C.c_int32),
        ("bch_m", C.c_int32), ("bch_t", C.c_int32), ("bch_prim", C.c_void_p),
        ("bps", C.c_int32), ("cstl", C.c_void_p),
        ("itl_cols", C.c_int32), ("itl_order", C.c_int32),
        ("fir_n_taps", C.c_int32), ("fir_taps", C.c_void_p), ("fir_osf", C.c_int32),
        ("max_frames", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p),
        ("ldpc_lds_groups", C.c_int32), ("pls", C.c_int32 * 7),
    ]


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N_ldpc", "K_ldpc", "K_bch", "bps", "N_xfec_sym", "pl_frame_sym", "ldpc_edges", "ldpc_q")]


# name -> (restype, argtypes).  Every symbol of include/dvbs2hip.h; tests/test_abi.py checks
# the list against the header and against the built .so.
SCHED_QC, SCHED_NATURAL = 0, 1
FIR_AUTO, FIR_VALU, FIR_MFMA = 0, 1, 2
_vp, _i, _f = C.c_void_p, C.c_int32, C.c_float
_SOCK2 = [_vp, _vp, _vp, _i]
ABI = {
    "dvbs2hip_cfg_from_modcod": (C.c_int, [C.c_char_p, C.POINTER(Cfg)]),
    "dvbs2hip_create": (C.c_int, [C.POINTER(Cfg), C.POINTER(_vp)]),
    "dvbs2hip_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "dvbs2hip_destroy": (None, [_vp]),
    "dvbs2hip_last_error": (C.c_char_p, [_vp]),
    "dvbs2hip_ldpc_kernel_name": (C.c_char_p, [_vp]),
    "dvbs2hip_set_ldpc_schedule": (C.c_int, [_vp, _i]),
    "dvbs2hip_set_filter_kernel": (C.c_int, [_vp, _i]),
    "dvbs2hip_host_register": (C.c_int, [_vp, _vp, C.c_size_t]),
    "dvbs2hip_host_unregister": (C.c_int, [_vp, _vp]),
    "dvbs2hip_reset": (C.c_int, [_vp]),
    "dvbs2hip_sync_lr_synchronize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_lr_synchronize_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_lr_set_alpha": (C.c_int, [_vp, _f]),
    "dvbs2hip_sync_lr_reset": (C.c_int, [_vp]),
    "dvbs2hip_sync_lr_timeouts": (C.c_int, [_vp, _vp]),
    "dvbs2hip_sync_freq_phase_synchronize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_freq_phase_synchronize_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_set_params": (C.c_int, [_vp, _f, _f, _i]),
    "dvbs2hip_sync_frame_reset": (C.c_int, [_vp]),
    "dvbs2hip_sync_frame_synchronize1": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_synchronize1_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_synchronize2": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_synchronize2_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_synchronize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_synchronize_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_locate_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_sync_frame_get_metric": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "dvbs2hip_set_ldpc_params": (C.c_int, [_vp, _i, _f, _i]),
    "dvbs2hip_get_stream": (_vp, [_vp]),
    "dvbs2hip_synchronize": (C.c_int, [_vp]),
    "dvbs2hip_graph_begin": (C.c_int, [_vp]),
    "dvbs2hip_graph_end": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "dvbs2hip_graph_launch": (C.c_int, [_vp, _i]),
    "dvbs2hip_graph_destroy": (C.c_int, [_vp, _i]),
    "dvbs2hip_get_sizes": (C.c_int, [_vp, C.POINTER(Sizes)]),
    "dvbs2hip_ldpc_decode_siho": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_ldpc_decode_siho_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_ldpc_decode_siho_post": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_bch_decode_hiho": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_bch_decode_hiho_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_demodulate": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_demodulate_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_deinterleave": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_deinterleave_dev": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_demodulate_deinterleave": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_demodulate_deinterleave_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_filter": (C.c_int, [_vp, _vp, _vp, _i, _i]),
    "dvbs2hip_filter_dev": (C.c_int, [_vp, _vp, _vp, _i, _i]),
    "dvbs2hip_filter_reset": (C.c_int, [_vp]),
    "dvbs2hip_filter_split": (C.c_int, [_vp, _i]),
    "dvbs2hip_filter1": (C.c_int, [_vp, _vp, _vp, _i, _i]),
    "dvbs2hip_filter1_dev": (C.c_int, [_vp, _vp, _vp, _i, _i]),
    "dvbs2hip_filter2": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i]),
    "dvbs2hip_filter2_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i]),
    "dvbs2hip_estimate": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_estimate_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_agc_imultiply": (C.c_int, [_vp, _vp, _vp, _i, C.c_float, _i]),
    "dvbs2hip_agc_imultiply_dev": (C.c_int, [_vp, _vp, _vp, _i, C.c_float, _i]),
    "dvbs2hip_sync_coarse_set_freq": (C.c_int, [_vp, C.c_float]),
    "dvbs2hip_sync_coarse_reset": (C.c_int, [_vp]),
    "dvbs2hip_sync_coarse_synchronize": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "dvbs2hip_sync_coarse_synchronize_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "dvbs2hip_pl_descramble": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_pl_descramble_dev": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_remove_plh": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_remove_plh_dev": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_bb_descramble": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_bb_descramble_dev": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_monitor_check_errors": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_monitor_check_errors_dev": (C.c_int, [_vp, _vp, _vp, _i]),
    "dvbs2hip_monitor_get": (C.c_int, [_vp, _vp]),
    "dvbs2hip_monitor_reset": (C.c_int, [_vp]),
    "dvbs2hip_monitor_check_errors2": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_monitor_check_errors2_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_monitor_reduce_init": (C.c_int, [_vp, _i, _i, C.c_char_p, _i]),
    "dvbs2hip_rendezvous": (C.c_int, [_i, _i, C.c_char_p, _vp, C.c_size_t, _i]),
    "dvbs2hip_monitor_reduce": (C.c_int, [_vp, _vp]),
    "dvbs2hip_monitor_reduce_finalize": (C.c_int, [_vp]),
    "dvbs2hip_rx_bb": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_rx_bb_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_rx_bb_located_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "dvbs2hip_tx_bb": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, _i]),
    "dvbs2hip_tx_bb_dev": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, _vp, _i]),
    "dvbs2hip_shape_filter": (C.c_int, [_vp, _vp, _vp, _i, _i]),
    "dvbs2hip_shape_filter_dev": (C.c_int, [_vp, _vp, _vp, _i, _i]),
    "dvbs2hip_add_noise": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, _i, _i]),
    "dvbs2hip_add_noise_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64, _i, _i]),
    "dvbs2hip_extract_dev": (C.c_int, [_vp, _vp, _vp, _i, _i, C.c_int64, _i]),
    "dvbs2hip_timing_enable": (C.c_int, [_vp, _i]),
    "dvbs2hip_timing_reset": (C.c_int, [_vp]),
    "dvbs2hip_timing_get": (C.c_int, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dvbs2hip_device_copy_bandwidth": (C.c_int, [_vp, C.c_size_t, _i, C.POINTER(C.c_double)]),
    "dvbs2hip_malloc": (C.c_int, [_vp, C.POINTER(_vp), C.c_size_t]),
    "dvbs2hip_free": (C.c_int, [_vp, _vp]),
    "dvbs2hip_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "dvbs2hip_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
}

K_LDPC, K_BCH, K_DEMAP, K_FIR, K_FRONT, K_MISC = range(6)


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """Loads libdvbs2hip.so (building it with hipcc when the in-tree copy is absent)."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm loads its OWN bundled libamdhip64 by absolute path.  If
        # libdvbs2hip.so (linked against /opt/rocm's) initialises HIP first, torch's copy then finds no device.
        # Importing torch first makes the dynamic loader resolve our DT_NEEDED libamdhip64 to the copy torch
        # already loaded.  (Hosts without PyTorch -- the C++ side -- are unaffected.)
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        path = os.environ.get("DVBS2HIP_LIB", _build.LIB)      # development: another build of the same sources (tools/build_variant.sh)
        if not os.path.exists(path):
            if not build_if_missing:
                raise Dvbs2HipError(-5, "libdvbs2hip.so not built (%s); there is no CPU fallback" % path)
            _build.build_lib()
        L = C.CDLL(path)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
