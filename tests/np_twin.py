"""An independent NumPy twin of the oracle's LDPC decoder ("two independent implementations agree"): H is
built here from the ETSI address table alone (EN 302 307 5.3.2: information bit 360 g + m accumulates into parity addresses
(x + m q) mod M for every x of table row g; parity bit k joins checks k and k + 1), the sweep is the horizontal-layered one in
NATURAL row order (what AFF3CT's Decoder_LDPC_BP_horizontal_layered does), the check node is normalised min-sum
with the value-equality rule, or the exact sum-product rule -- in the oracle's form (forward / backward boxplus recursions) and in
the form the GPU kernel evaluates (complement products, k_ldpc_wg8.hip).  fp32 throughout.  Test infrastructure only."""
import numpy as np

f32 = np.float32
LOG2E = f32(1.4426950408889634)
LN2 = f32(0.6931471805599453)


def build_checks(N, K, row_ptr, addr):
    """-> list of M int arrays: the variables of check k, information edges in table order, then p_k, then p_{k-1}."""
    M = N - K
    q = M // 360
    chk = [[] for _ in range(M)]
    for g in range(len(row_ptr) - 1):
        for p in range(row_ptr[g], row_ptr[g + 1]):
            for m in range(360):
                chk[(int(addr[p]) + m * q) % M].append(g * 360 + m)
    for k in range(M):
        chk[k].append(K + k)
        if k > 0:
            chk[k].append(K + k - 1)
    return [np.asarray(c, dtype=np.int64) for c in chk]


def nms_check(v2c, alpha):
    a = np.abs(v2c)
    order = np.sort(a)
    min1, min2 = order[0], order[1]
    tot = np.logical_xor.reduce(np.signbit(v2c))
    mag = np.where(a == min1, f32(min2 * f32(alpha)), f32(min1 * f32(alpha))).astype(f32)
    neg = tot ^ np.signbit(v2c)
    return np.where(neg, -mag, mag).astype(f32)


def boxplus(a, b):
    a = np.asarray(a, f32); b = np.asarray(b, f32)
    mn = np.minimum(np.abs(a), np.abs(b))
    sg = np.where(np.signbit(a) ^ np.signbit(b), -mn, mn).astype(f32)
    with np.errstate(all="ignore"):
        r = (sg + (np.log1p(np.exp(-np.abs(a + b), dtype=f32), dtype=f32) - np.log1p(np.exp(-np.abs(a - b), dtype=f32), dtype=f32))).astype(f32)
    return np.where(np.isinf(a), b, np.where(np.isinf(b), a, r)).astype(f32)


def spa_check_boxplus(v2c):
    """one check (1-D) or a batch [n, d]: out_j = boxplus of all the other inputs, forward / backward recursions (oracle form)"""
    x = np.atleast_2d(np.asarray(v2c, f32))
    n, d = x.shape
    fw = np.empty((n, d), f32); bw = np.empty((n, d), f32)
    fw[:, 0] = np.inf
    for j in range(1, d):
        fw[:, j] = boxplus(fw[:, j - 1], x[:, j - 1])
    bw[:, d - 1] = np.inf
    for j in range(d - 2, -1, -1):
        bw[:, j] = boxplus(bw[:, j + 1], x[:, j + 1])
    out = boxplus(fw, bw)
    return out[0] if np.ndim(v2c) == 1 else out


def spa_check_complement(v2c):
    """The same check node as the GPU kernel evaluates it (k_ldpc_wg8.hip, SPA layer): Q' = 2^s2 (1 - prod tanh(a_i / 2)) by prefix /
    suffix recursions of positive terms, |out| = ln((2 - Q) / Q); s2 from the second smallest magnitude keeps every sum in range."""
    x = np.atleast_2d(np.asarray(v2c, f32))
    n, d = x.shape
    a = np.abs(x)
    srt = np.sort(a, axis=1)
    min1, min2 = srt[:, 0], srt[:, 1]
    with np.errstate(all="ignore"):
        s2 = np.maximum(f32(0), (min2 - f32(16)) * LOG2E).astype(f32)
        kap = np.exp2(-s2, dtype=f32)
        cln = (s2 * LN2).astype(f32)
        hk = (f32(0.5) * kap).astype(f32)
        # u' = 2^s2 . 2 / (e^a + 1) = 1 / (2^(a log2 e - s2 - 1) + 2^-(s2 + 1))
        u = (f32(1) / (np.exp2((a * LOG2E - (s2 + f32(1))[:, None]).astype(f32), dtype=f32) + hk[:, None]).astype(f32)).astype(f32)
        comb = lambda A, b: (b * (f32(1) - kap * A).astype(f32) + A).astype(f32)
        B = np.zeros((n, d), f32)
        for j in range(d - 2, -1, -1):
            B[:, j] = comb(B[:, j + 1], u[:, j + 1])
        key = np.where(min2 - min1 > f32(60), min1, f32(np.nan)).astype(f32)
        A = np.zeros(n, f32)
        out = np.empty((n, d), f32)
        for j in range(d):
            Q = comb(A, B[:, j])
            lg = (np.log2((f32(2) - kap * Q).astype(f32), dtype=f32) - np.log2(Q, dtype=f32)).astype(f32)
            o = (lg * LN2 + cln).astype(f32)
            out[:, j] = np.where((a[:, j] < key) | (a[:, j] > key), min1, o)
            A = comb(A, u[:, j])
    tot = np.logical_xor.reduce(np.signbit(x), axis=1)
    neg = tot[:, None] ^ np.signbit(x)
    out = np.where(neg, -np.abs(out), np.abs(out)).astype(f32)
    return out[0] if np.ndim(v2c) == 1 else out


def decode_natural(chk, llr, K, n_ite, alpha=1.0, implem="NMS", early_stop=False):
    """horizontal-layered sweep over the checks in natural order; -> bits[K], posteriors[N], cwd, iterations done"""
    L = np.asarray(llr, f32).copy()
    msg = [np.zeros(len(c), f32) for c in chk]
    upd = (lambda v: nms_check(v, alpha)) if implem == "NMS" else spa_check_boxplus
    synd_ok = lambda: all(not (np.count_nonzero(L[c] < 0) & 1) for c in chk)
    it = 0
    while it < n_ite:
        for k, c in enumerate(chk):
            v2c = (L[c] - msg[k]).astype(f32)
            nw = upd(v2c)
            msg[k] = nw
            L[c] = (v2c + nw).astype(f32)
        it += 1
        if early_stop and synd_ok():
            break
    return (L[:K] < 0).astype(np.int32), L, int(synd_ok()), it
