"""Shared test helpers: seeded synthetic frames made with the CPU oracle's TX chain."""
import numpy as np

from dvbs2_amd import params as P

_chains = {}


def chain(O, modcod):
    if modcod not in _chains:
        _chains[modcod] = O.Chain(P.get_modcod(modcod))
    return _chains[modcod]


def sigma_for(mc, ebn0_db):
    return P.esn0_to_sigma(P.ebn0_to_esn0(ebn0_db, mc.code_rate, mc.bps))


def make_pl_frames(O, modcod, F, ebn0_db, seed):
    """-> info[F,K_bch] int32, pl_frames[F, 2*pl_frame] f32 (noisy), cw[F,N] int32, sigma"""
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(seed)
    sigma = sigma_for(mc, ebn0_db)
    info = rng.integers(0, 2, (F, mc.K_bch)).astype(np.int32)
    pl = np.empty((F, 2 * mc.pl_frame), dtype=np.float32)
    cws = np.empty((F, mc.N_ldpc), dtype=np.int32)
    for f in range(F):
        plf, cw = ch.tx(info[f])
        pl[f] = plf + (sigma * rng.standard_normal(plf.size)).astype(np.float32)
        cws[f] = cw
    return info, pl, cws, sigma


def make_llrs(O, modcod, F, ebn0_db, seed):
    """BPSK-equivalent channel LLRs for LDPC-only tests (BASELINE config 2):
    y = (1-2c) + sigma n, LLR = 2 y / sigma^2.  -> info_ldpc[F,K], llr[F,N], cw[F,N]"""
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(seed)
    rate = mc.K_bch / mc.N_ldpc
    sigma = float(np.sqrt(1.0 / (2.0 * rate * 10.0 ** (ebn0_db / 10.0))))
    info = rng.integers(0, 2, (F, mc.K_bch)).astype(np.int32)
    bch = ch.bch.encode(info)
    cw = ch.ldpc.encode(bch)
    y = (1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.shape)
    llr = (2.0 * y / sigma ** 2).astype(np.float32)
    return bch, llr, cw
