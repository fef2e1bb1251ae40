"""The C-ABI library builds, loads and exports every symbol include/dvbs2hip.h declares; the
Python binding covers exactly that list; without a GPU the product fails loudly (no fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dvbs2hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dvbs2hip_[a-z0-9_]+)\s*\(", txt)))


def test_header_binding_and_library_agree():
    from dvbs2_amd import build, lib_binding
    lib = build.build_lib()
    syms = header_symbols()
    assert len(syms) >= 40
    assert sorted(lib_binding.ABI) == syms
    nm = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (dvbs2hip_\w+)", nm))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    L = lib_binding.load()
    for s in syms:
        assert hasattr(L, s)
    # torch types never appear in the ABI: the library does not link libtorch
    ldd = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
    assert "libtorch" not in ldd and "libc10" not in ldd


def test_cfg_from_modcod_and_errors_without_gpu():
    from dvbs2_amd import lib_binding as B
    L = B.load()
    cfg = B.Cfg()
    assert L.dvbs2hip_cfg_from_modcod(b"16APSK-S_8/9", ctypes.byref(cfg)) == 0
    assert (cfg.N_ldpc, cfg.K_ldpc, cfg.K_bch, cfg.bps, cfg.itl_cols, cfg.bch_m, cfg.bch_t) == (16200, 14400, 14232, 4, 4, 14, 12)
    assert cfg.ldpc_n_ite == 50 and cfg.ldpc_implem == 2 and cfg.ldpc_alpha == 1.0 and cfg.fir_n_taps == 81       # reference defaults: SPA (2), 50 ite (DVBS2.cpp:135-138)
    taps = np.ctypeslib.as_array(ctypes.cast(cfg.fir_taps, ctypes.POINTER(ctypes.c_float)), (81,))
    from dvbs2_amd import params as P
    assert np.array_equal(taps, P.rrc_taps(0.2, 2, 20))
    assert L.dvbs2hip_cfg_from_modcod(b"", ctypes.byref(cfg)) == 0 and cfg.K_bch == 14232   # DVBS2.cpp:290
    assert L.dvbs2hip_cfg_from_modcod(b"QAM-1/2", ctypes.byref(cfg)) == -1
    assert b"mod-cod scheme not supported" in L.dvbs2hip_last_error(None)                    # DVBS2.cpp:319
    import torch
    if not torch.cuda.is_available():
        # CPU-only container: creating a handle must FAIL, never fall back to host compute
        L.dvbs2hip_cfg_from_modcod(b"QPSK-S_8/9", ctypes.byref(cfg))
        h = ctypes.c_void_p()
        rc = L.dvbs2hip_create(ctypes.byref(cfg), ctypes.byref(h))
        assert rc == -5 and not h.value
        assert b"no CPU fallback" in L.dvbs2hip_last_error(None)
        from dvbs2_amd.receiver import Dvbs2Hip
        with pytest.raises(B.Dvbs2HipError):
            Dvbs2Hip("QPSK-S_8/9")


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under dvbs2_amd/, include/ or the C++ host
    side may import, include, link or call it."""
    import ast
    bad = []
    for base in ("dvbs2_amd", "include", "host"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                path = os.path.join(dp, f)
                if f.endswith(".py"):
                    for node in ast.walk(ast.parse(open(path).read())):
                        names = []
                        if isinstance(node, ast.Import):
                            names = [a.name for a in node.names]
                        elif isinstance(node, ast.ImportFrom):
                            names = [node.module or ""]
                        if any(n.split(".")[0] == "oracle" for n in names):
                            bad.append((path, names))
                    if re.search(r"libdvbs2_oracle|\borc_\w+\(", open(path).read()):
                        bad.append((path, "oracle symbol"))
                elif f.endswith((".hip", ".h", ".hpp", ".cpp", ".c")):
                    txt = open(path, errors="replace").read()
                    if re.search(r'#\s*include\s*[<"][^>"]*oracle|libdvbs2_oracle|\borc_\w+\s*\(', txt):
                        bad.append((path, "oracle include/symbol"))
    assert not bad, bad
    from dvbs2_amd import build
    ldd = subprocess.run(["ldd", build.LIB], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
