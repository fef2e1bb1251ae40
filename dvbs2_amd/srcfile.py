"""The reference's payload sources and its sink as files (DVBS2.cpp:359-389; the modules themselves live in StreamPU, absent from /root/reference):

  Source_user         conf/src/K_14232.src, K_9552.src: text, `n_frames`, `K`, then n_frames * K bits; the source cycles through the frames (DVBS2.cpp:367)
  Source_user_binary  `--src-type USER_BIN --src-path video.ts` (README.md:203): any file, eight payload bits per byte, the file's first bit in the least significant
                      bit of its first byte [UPSTREAM-RECALL: spu::tools::Bit_packer's default order]; at the end of the file it starts over (`--src-no-loop`: the last
                      frame is zero-padded and the source is done) (DVBS2.cpp:369)
  Source_AZCW         all-zero payloads (DVBS2.cpp:371)
  Sink_user_binary    `--snk-path out.ts` (README.md:210): the decoded payload packed the same way, frame after frame (DVBS2.cpp:385)

A file sent with USER_BIN and received into the sink is the file again, byte for byte, whatever the bit order -- both ends use the same one."""
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


class SourceDone(RuntimeError):
    """the source has no more frames (a USER_BIN file without looping has been sent)"""


class SourceUserBinary:
    """generate(F) -> the next F frames of K bits from a binary file, int32 [F, K]"""

    def __init__(self, path: str, K: int, auto_reset: bool = True):
        if K % 8:
            raise ValueError("K = %d is not a whole number of bytes" % K)
        import os
        if os.path.getsize(path) == 0:
            raise ValueError("'%s' is empty" % path)
        self.data = np.memmap(path, dtype=np.uint8, mode="r")          # (a transport stream of gigabytes stays on disk)
        self.K, self.pos, self.auto_reset, self.done = K, 0, auto_reset, False

    def generate(self, F: int) -> np.ndarray:
        if self.done:
            raise SourceDone("the whole file has been sent")
        nb = F * self.K // 8
        if self.auto_reset:
            idx = (self.pos + np.arange(nb)) % self.data.size
            chunk = np.asarray(self.data[idx])
            self.pos = int((self.pos + nb) % self.data.size)
        else:
            chunk = np.zeros(nb, np.uint8)
            left = self.data[self.pos:self.pos + nb]
            chunk[:left.size] = left
            self.pos += left.size
            if self.pos >= self.data.size:
                self.done = True
        return np.unpackbits(chunk, bitorder="little").astype(np.int32).reshape(F, self.K)

    def frames_left(self) -> int | None:
        """whole or partial frames still to come (None: a looping source never ends)"""
        return None if self.auto_reset else -(-(self.data.size - self.pos) * 8 // self.K)


class SourceAZCW:
    def __init__(self, K: int):
        self.K = K

    def generate(self, F: int) -> np.ndarray:
        return np.zeros((F, self.K), np.int32)


class SinkUserBinary:
    """send(bits[F, K]) appends the frames to the file, eight bits per byte"""

    def __init__(self, path: str, K: int):
        if K % 8:
            raise ValueError("K = %d is not a whole number of bytes" % K)
        self.K, self.f = K, open(path, "wb")

    def send(self, bits) -> None:
        b = np.ascontiguousarray(bits).reshape(-1, self.K)
        self.f.write(np.packbits((b != 0).astype(np.uint8), axis=1, bitorder="little").tobytes())
        self.f.flush()                                               # (the sink may be a FIFO with a player behind it: README.md:206-217 of the reference)

    def close(self) -> None:
        self.f.close()
