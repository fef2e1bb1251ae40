"""MODCOD parameter table for the DVB-S2 RX inner path.

Restates ``factory::DVBS2::modcod_init`` (/root/reference src/common/Factory/DVBS2/DVBS2.cpp:287-356)
as plain data, extended with the normal-frame / 32APSK rows BASELINE.json's configs 2-5 need
(ETSI EN 302 307; not supported by the reference, whose N_ldpc is the constant 16200:
DVBS2.hpp:49).  Pure host logic: no GPU, no oracle.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

PL_M = 90    # slot length / PLHEADER length in symbols   (Framer.hxx:38-40)
PL_P = 36    # pilot block length in symbols

# BCH primitive polynomials, coefficient of x^i at index i
BCH_PRIM_SHORT = [1, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1]          # DVBS2.hpp:55, GF(2^14)
BCH_PRIM_NORMAL = [1, 0, 1, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]   # ETSI 5.3.1, GF(2^16)


@dataclass(frozen=True)
class ModCod:
    name: str
    mod: str
    cod: str
    bps: int
    cstl_file: str
    N_ldpc: int
    K_ldpc: int           # = N_bch
    K_bch: int
    bch_m: int
    bch_t: int
    itl_cols: int         # 1 = no interleaver
    itl_order: int        # 0 TOP_LEFT, 1 TOP_RIGHT
    pls: tuple            # 7-entry vector of Framer.hxx:111-126
    in_reference: bool
    ldpc_table: str

    @property
    def N_bch(self) -> int:
        return self.K_ldpc

    @property
    def N_xfec(self) -> int:          # symbols
        return self.N_ldpc // self.bps

    @property
    def S(self) -> int:
        return self.N_xfec // PL_M

    @property
    def N_pilots(self) -> int:
        # the reference's formula (DVBS2.cpp:353); equals the standard's for every row here
        return self.N_xfec // (16 * PL_M)

    @property
    def pl_frame(self) -> int:        # symbols
        return PL_M * (self.S + 1) + self.N_pilots * PL_P

    @property
    def code_rate(self) -> float:     # R used for SNR conversion (TX_RX_BB/main.cpp:142)
        return self.K_bch / self.N_ldpc

    @property
    def bch_prim(self) -> List[int]:
        return BCH_PRIM_SHORT if self.bch_m == 14 else BCH_PRIM_NORMAL


def _mc(name, mod, cod, bps, cstl, N, Kl, Kb, m, t, cols, order, pls, ref, table):
    return ModCod(name, mod, cod, bps, cstl, N, Kl, Kb, m, t, cols, order, tuple(pls), ref, table)


MODCODS = {m.name: m for m in [
    # ---- the five MODCODs the reference accepts (DVBS2.cpp:287-356, Framer.hxx:111-126)
    _mc("QPSK-S_8/9",   "QPSK",   "8/9", 2, "4QAM_GRAY.mod", 16200, 14400, 14232, 14, 12, 1, 0, [0, 0, 1, 0, 1, 0, 1], True,  "N16200_8_9.txt"),
    _mc("QPSK-S_3/5",   "QPSK",   "3/5", 2, "4QAM_GRAY.mod", 16200,  9720,  9552, 14, 12, 1, 0, [0, 0, 0, 1, 0, 1, 1], True,  "N16200_3_5.txt"),
    _mc("8PSK-S_3/5",   "8PSK",   "3/5", 3, "8PSK.mod",      16200,  9720,  9552, 14, 12, 3, 1, [0, 0, 1, 1, 0, 0, 1], True,  "N16200_3_5.txt"),
    _mc("8PSK-S_8/9",   "8PSK",   "8/9", 3, "8PSK.mod",      16200, 14400, 14232, 14, 12, 3, 0, [0, 1, 0, 0, 0, 0, 1], True,  "N16200_8_9.txt"),
    _mc("16APSK-S_8/9", "16APSK", "8/9", 4, "16APSK.mod",    16200, 14400, 14232, 14, 12, 4, 0, [0, 1, 0, 1, 1, 0, 1], True,  "N16200_8_9.txt"),
    # ---- extensions required by BASELINE.json configs 2-5 (ETSI EN 302 307)
    _mc("QPSK-N_8/9",   "QPSK",   "8/9", 2, "4QAM_GRAY.mod", 64800, 57600, 57472, 16,  8, 1, 0, [0, 0, 1, 0, 1, 0, 0], False, "N64800_8_9.txt"),
    _mc("8PSK-N_8/9",   "8PSK",   "8/9", 3, "8PSK.mod",      64800, 57600, 57472, 16,  8, 3, 0, [0, 1, 0, 0, 0, 0, 0], False, "N64800_8_9.txt"),
    _mc("16APSK-N_8/9", "16APSK", "8/9", 4, "16APSK.mod",    64800, 57600, 57472, 16,  8, 4, 0, [0, 1, 0, 1, 1, 0, 0], False, "N64800_8_9.txt"),
    _mc("32APSK-S_3/4", "32APSK", "3/4", 5, "32APSK_3_4.mod", 16200, 11880, 11712, 14, 12, 5, 0, [0, 1, 1, 0, 0, 0, 1], False, "N16200_3_4.txt"),
]}


def get_modcod(name: str) -> ModCod:
    if name == "":
        name = "QPSK-S_8/9"        # DVBS2.cpp:290 default
    if name not in MODCODS:
        # same failure mode as DVBS2.cpp:319
        raise ValueError(name + " mod-cod scheme not supported.")
    return MODCODS[name]


def load_ldpc_table(fname: str):
    """-> (row_ptr int32[n_rows+1], addr int32[n_addr]) from dvbs2_amd/data/ldpc/<fname>."""
    rows = []
    with open(os.path.join(_DATA, "ldpc", fname)) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            rows.append([int(x) for x in line.split()])
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int32)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    addr = np.array([a for r in rows for a in r], dtype=np.int32)
    return row_ptr, addr


def load_constellation(fname: str) -> np.ndarray:
    """Raw (un-normalised) points, float32 [n_pts, 2]; index = line order (conf/mod/*.mod)."""
    pts = []
    with open(os.path.join(_DATA, "mod", fname)) as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            a, b = line.split()[:2]
            pts.append((float(a), float(b)))
    return np.array(pts, dtype=np.float32)


def normalise_constellation(pts: np.ndarray) -> np.ndarray:
    """Unit mean energy in fp32, as tools::Constellation_user does."""
    pts = np.asarray(pts, dtype=np.float32)
    es = np.float32(0)
    for p in pts:
        es = np.float32(es + np.float32(p[0] * p[0] + p[1] * p[1]))
    s = np.sqrt(np.float32(es / np.float32(len(pts))), dtype=np.float32)
    return (pts / s).astype(np.float32)


def ebn0_to_esn0(ebn0_db: float, rate: float, bps: int) -> float:
    return ebn0_db + 10.0 * math.log10(rate * bps)     # TX_RX_BB/main.cpp:142-146


def esn0_to_sigma(esn0_db: float) -> float:
    return math.sqrt(1.0 / (2.0 * 10.0 ** (esn0_db / 10.0)))


def rrc_taps(rolloff: float = 0.2, osf: int = 2, grp_delay: int = 20) -> np.ndarray:
    """fp32 restatement of Filter_RRC_ccr_naive::compute_rrc_coefs
    (src/common/Module/Filter/Filter_FIR/Filter_RRC/Filter_RRC_ccr_naive.cpp:13-48);
    defaults from Shaping_filter.hpp:24-28 -> 81 taps."""
    f = np.float32
    pi = f(3.1415926535897932384626433832795)
    ro = f(rolloff)
    c = grp_delay * osf
    taps = np.zeros(2 * c + 1, dtype=np.float32)
    eps = np.finfo(np.float32).eps
    taps[c] = f(1.0) - ro + f(4.0) * ro / pi
    en = f(taps[c] * taps[c])
    for i in range(1, c + 1):
        t = f(i) / f(osf)
        if abs(f(4.0) * ro * t - f(1.0)) <= eps or abs(f(4.0) * ro * t + f(1.0)) <= eps:
            v = ro / np.sqrt(f(2.0)) * ((f(1.0) + f(2.0) / pi) * np.sin(pi / (f(4.0) * ro)) +
                                         (f(1.0) - f(2.0) / pi) * np.cos(pi / (f(4.0) * ro)))
        else:
            den = pi * t * (f(1.0) - f(16.0) * ro * ro * t * t)
            num = np.sin(pi * t * (f(1.0) - ro)) + f(4.0) * ro * t * np.cos(pi * t * (f(1.0) + ro))
            v = num / den
        taps[c + i] = v
        taps[c - i] = v
        en = f(en + f(v * v + v * v))
    return (taps / np.sqrt(en)).astype(np.float32)
