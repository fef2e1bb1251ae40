"""N3 -- the reference's raw IQ wire format (Radio_user_binary): native-endian interleaved re/im
samples of type R in {float64, float32, int16, int8}, 2 * N * sizeof(R) bytes per frame, no header
(/root/reference src/common/Module/Radio/Radio_user/Radio_user_binary.cpp:55-121).  This is what
`dvbs2_tx --rad-tx-file-path` / `dvbs2_ch` write and `dvbs2_rx --rad-rx-file-path` reads
(README.md:151-169), so the GPU receiver can consume the same files.

Semantics mirrored: `receive()` fills n_frames frames; at end of file it rewinds when `auto_reset`
(the default, :98-99) else sets `done` and raises ProcessingAborted (:101-102); `send()` appends.
"""
from __future__ import annotations

import numpy as np

DTYPES = {"f64": np.float64, "f32": np.float32, "i16": np.int16, "i8": np.int8,
          "double": np.float64, "float": np.float32, "int16": np.int16, "int8": np.int8}


class ProcessingAborted(RuntimeError):
    """spu::tools::processing_aborted: cooperative end of a sequence iteration."""


class RadioUserBinary:
    def __init__(self, N: int, input_filename: str = "", output_filename: str = "", auto_reset: bool = True,
                 n_frames: int = 1, dtype: str = "f32"):
        if N <= 0:
            raise ValueError("'N' has to be greater than 0")
        if dtype not in DTYPES:
            raise ValueError("unsupported sample type '%s'" % dtype)
        self.N, self.n_frames, self.auto_reset, self.done = int(N), int(n_frames), bool(auto_reset), False
        self.dtype = np.dtype(DTYPES[dtype])
        self._in = self._out = None
        try:
            if input_filename:
                self._in = open(input_filename, "rb")
            if output_filename:
                self._out = open(output_filename, "wb")
        except OSError:
            raise RuntimeError("'%s' file name is invalid: failbit is set." % (input_filename or output_filename))

    @property
    def frame_bytes(self) -> int:
        return 2 * self.N * self.dtype.itemsize

    def is_done(self) -> bool:
        return self.done

    def reset(self):
        if self._in:
            self._in.seek(0)

    def receive(self) -> np.ndarray:
        """-> array [n_frames, 2*N] of the file's sample type"""
        if self._in is None:
            raise RuntimeError("'input_file' is not open.")
        out = np.empty((self.n_frames, 2 * self.N), dtype=self.dtype)
        for f in range(self.n_frames):
            buf = self._in.read(self.frame_bytes)
            if len(buf) < self.frame_bytes:
                if not self.auto_reset:
                    self.done = True
                    raise ProcessingAborted()
                self.reset()
                buf = self._in.read(self.frame_bytes)
                if len(buf) < self.frame_bytes:
                    raise RuntimeError("Unknown error during file reading.")
            out[f] = np.frombuffer(buf, dtype=self.dtype)
        return out

    def send(self, X_N1):
        if self._out is None:
            raise RuntimeError("'output_file' is not open.")
        x = np.ascontiguousarray(X_N1, dtype=self.dtype)
        if x.size % (2 * self.N):
            raise ValueError("socket size is not a multiple of 2 * N")
        self._out.write(x.tobytes())
        self._out.flush()

    def close(self):
        for fh in (self._in, self._out):
            if fh:
                fh.close()
        self._in = self._out = None
