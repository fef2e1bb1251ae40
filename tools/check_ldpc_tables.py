#!/usr/bin/env python3
"""Structural validation of the DVB-S2 LDPC address tables.

The tables are entered from ETSI EN 302 307 Annex B/C (the reference keeps them in
the absent lib/aff3ct).  This script asserts the invariants a correct table must
satisfy; run it after touching any table.
"""
import sys, os, collections
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
DIR = os.path.join(HERE, "..", "dvbs2_amd", "data", "ldpc")

def load(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        rows.append([int(x) for x in line.split()])
    return rows

def check(name, N, K):
    rows = load(os.path.join(DIR, name))
    M = N - K
    q = M // 360
    assert M % 360 == 0 and K % 360 == 0
    assert len(rows) == K // 360, (name, len(rows), K // 360)
    flat = [a for r in rows for a in r]
    assert all(0 <= a < M for a in flat), name
    # uniform check degree <=> every residue class mod q holds the same number of addresses
    res = collections.Counter(a % q for a in flat)
    per = len(flat) // q
    bad = {r: c for r, c in res.items() if c != per}
    # no duplicate edges: within one row no two addresses equal
    dup = [i for i, r in enumerate(rows) if len(set(r)) != len(r)]
    # 4-cycles inside the info part: two bit-groups sharing two checks
    # build edges
    E = len(flat) * 360 + 2 * M - 1
    deg = collections.Counter(len(r) for r in rows)
    # same-layer conflicts (two addresses of one row in the same residue class)
    conf = sum(1 for r in rows if len(set(a % q for a in r)) != len(r))
    # 4-cycle count via sparse H^T H on info part
    cols = []
    for g, r in enumerate(rows):
        for m in range(360):
            cols.append(sorted((a + m * q) % M for a in r))
    from scipy.sparse import csr_matrix
    ii = np.concatenate([np.full(len(c), i) for i, c in enumerate(cols)])
    jj = np.concatenate([np.array(c) for c in cols])
    # add parity columns
    pi = []; pj = []
    for c in range(M):
        pi += [K + c]; pj += [c]
        if c + 1 < M:
            pi += [K + c]; pj += [c + 1]
    ii = np.concatenate([ii, np.array(pi)]); jj = np.concatenate([jj, np.array(pj)])
    H = csr_matrix((np.ones(len(ii), dtype=np.int32), (jj, ii)), shape=(M, N))
    assert H.nnz == E, (H.nnz, E)
    rowdeg = np.asarray(H.sum(axis=1)).ravel()
    G = (H @ H.T).tocoo()
    off = G.data[G.row != G.col]
    n4 = int(((off * (off - 1)) // 2).sum() // 2)
    print(f"{name}: rows={len(rows)} q={q} E={E} deg_hist={dict(deg)} per_residue={per} "
          f"bad_residues={bad} dup_rows={dup} same_layer_conflict_rows={conf} "
          f"row_weight_hist={dict(collections.Counter(rowdeg.tolist()))} four_cycles={n4}")
    return not bad and not dup

ok = True
ok &= check("N16200_8_9.txt", 16200, 14400)
ok &= check("N16200_3_5.txt", 16200, 9720)
ok &= check("N16200_3_4.txt", 16200, 11880)
ok &= check("N64800_8_9.txt", 64800, 57600)
sys.exit(0 if ok else 1)
