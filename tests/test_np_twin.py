"""The oracle's self-check SURVEY.md 8(c) asks for: a second, independent implementation (NumPy, tests/np_twin.py -- its own H from
the ETSI address table, its own sweep and check-node rules) against the C oracle: NMS bit for bit, SPA within 1e-5; and the form of the
sum-product check node the GPU kernel evaluates (complement products) against the oracle's boxplus recursions over the whole LLR range."""
import numpy as np
import pytest

import np_twin as T
from helpers import chain, make_llrs


@pytest.mark.parametrize("modcod,ebn0", [("QPSK-S_8/9", 4.0), ("QPSK-S_8/9", 3.0), ("QPSK-S_3/5", 1.4), ("32APSK-S_3/4", 3.0)])
def test_numpy_twin_equals_the_c_oracle(O, P, modcod, ebn0):
    ch = chain(O, modcod)
    mc = ch.mc
    row_ptr, addr = P.load_ldpc_table(mc.ldpc_table)
    chk = T.build_checks(mc.N_ldpc, mc.K_ldpc, row_ptr, addr)
    # the twin's H against the oracle's (as sets of variables per check) and the edge count of SURVEY 8(d)
    cp, cv = ch.ldpc.csr()
    assert len(chk) == mc.N_ldpc - mc.K_ldpc and sum(len(c) for c in chk) == cp[-1]
    for k in (0, 1, 2, 357, len(chk) - 1):
        assert sorted(chk[k].tolist()) == sorted(cv[cp[k]:cp[k + 1]].tolist())
    _, llr, cw = make_llrs(O, modcod, 1, ebn0, seed=41)
    for n_ite, es in ((3, False), (10, True)):
        b, post, cwd, it = T.decode_natural(chk, llr[0], mc.K_ldpc, n_ite, alpha=0.875, implem="NMS", early_stop=es)
        bo, po, co, io = ch.ldpc.decode(llr, n_ite=n_ite, alpha=0.875, implem=O.NMS, sched=O.NATURAL, early_stop=es)
        assert np.array_equal(b, bo[0]) and cwd == co[0] and it == io[0]
        assert np.array_equal(post, po[0]), "NMS: the twin and the oracle are expected to agree bit for bit"
    b, post, cwd, it = T.decode_natural(chk, llr[0], mc.K_ldpc, 2, implem="SPA")
    bo, po, co, io = ch.ldpc.decode(llr, n_ite=2, implem=O.SPA, sched=O.NATURAL, early_stop=False)
    assert np.max(np.abs(post - po[0]) / np.maximum(1.0, np.abs(po[0]))) <= 1e-5
    assert (b != bo[0]).mean() < 1e-3


def test_complement_product_spa_equals_the_boxplus_recursions():
    """k_ldpc_wg8.hip's sum-product check node (NumPy restatement of its arithmetic) against the oracle's form, far beyond the range
    where tanh saturates: LLRs of a few hundred, one weak edge among strong ones, zeros, ties -- the bar is 1e-4 max(1, |out|), met
    with more than a decade to spare."""
    rng = np.random.default_rng(1)
    n = 4000
    worst = 0.0
    for d in (27, 13, 11, 4):
        cases = [rng.normal(5, 3, (n, d)), rng.normal(0, 1, (n, d)), rng.normal(20, 6, (n, d)), rng.normal(100, 30, (n, d)), rng.normal(400, 100, (n, d)),
                 rng.normal(0, 1, (n, d)) * np.exp(rng.uniform(-8, 7, (n, d)))]
        x = rng.normal(150, 30, (n, d)); x[:, 3] = rng.normal(0, 2, n); cases.append(x)                       # one weak edge among strong ones
        x = rng.normal(150, 30, (n, d)); x[:, 3] = rng.normal(0, 2, n); x[:, 1] = rng.normal(40, 10, n); cases.append(x)
        x = rng.normal(300, 30, (n, d)); x[:, 2] = rng.normal(100, 5, n); cases.append(x)                     # the weakest edge overflows the scaled range
        x = rng.normal(5, 3, (n, d)); x[:, 2] = 0.0; x[:, 0] = -0.0; cases.append(x)
        x = rng.normal(30, 3, (n, d)); x[:, 1] = x[:, 2]; cases.append(x)                                     # ties at any rank
        x = rng.normal(5, 3, (n, d)); x[:, d - 1] = np.inf; cases.append(x)                                   # the neutral element (absent edge, NULL slots)
        for x in cases:
            x = x.astype(np.float32)
            o = T.spa_check_boxplus(x); g = T.spa_check_complement(x)
            fin = np.isfinite(x)
            assert not np.isnan(g[fin]).any()
            assert np.array_equal(np.signbit(o[fin]), np.signbit(g[fin])) or np.all(np.abs(o[fin][np.signbit(o[fin]) != np.signbit(g[fin])]) < 1e-6)
            err = np.abs(o - g)[fin] / np.maximum(1.0, np.abs(o[fin]))
            worst = max(worst, float(err.max()))
    assert worst <= 1e-5, worst
