"""Source_user pattern files of the reference (conf/src/K_14232.src, K_9552.src): text, `n_frames`, `K`, then
n_frames * K bits (aff3ct Source_user, built at DVBS2.cpp:367).  The source cycles through the frames of the file."""
from __future__ import annotations

import numpy as np


def load_src(path: str, K: int | None = None) -> np.ndarray:
    """-> int32 [n_frames, K]"""
    tok = open(path).read().split()
    if len(tok) < 2:
        raise ValueError("'%s' is not a source pattern file" % path)
    n, k = int(tok[0]), int(tok[1])
    if K is not None and k != K:
        raise ValueError("'%s' holds frames of %d bits, the MODCOD needs %d" % (path, k, K))
    bits = np.array(tok[2:2 + n * k], dtype=np.int32)
    if bits.size != n * k or ((bits != 0) & (bits != 1)).any():
        raise ValueError("'%s' is truncated or holds something else than bits" % path)
    return bits.reshape(n, k)


def save_src(path: str, bits) -> None:
    b = np.atleast_2d(np.asarray(bits, dtype=np.int32))
    with open(path, "w") as f:
        f.write("%d\n%d\n" % b.shape)
        for row in b:
            f.write(" ".join(map(str, row.tolist())) + " \n")


class SourceUser:
    """generate(F) -> the next F frames of the pattern, cyclically (auto_reset = true)"""

    def __init__(self, path: str, K: int):
        self.frames = load_src(path, K)
        self.pos = 0

    def generate(self, F: int) -> np.ndarray:
        idx = (self.pos + np.arange(F)) % self.frames.shape[0]
        self.pos = int((self.pos + F) % self.frames.shape[0])
        return self.frames[idx]
