"""N3: the reference's Radio_user_binary file format (host logic, runs on CPU)."""
import numpy as np
import pytest

from dvbs2_amd.iqfile import ProcessingAborted, RadioUserBinary


@pytest.mark.parametrize("dtype,npdt", [("f32", np.float32), ("f64", np.float64), ("i16", np.int16), ("i8", np.int8)])
def test_roundtrip_and_frame_size(tmp_path, dtype, npdt):
    N, F = 33, 3
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((5, 2 * N)) * 20).astype(npdt)
    path = str(tmp_path / "iq.bin")
    tx = RadioUserBinary(N, output_filename=path, dtype=dtype)
    tx.send(x)
    tx.close()
    import os
    assert os.path.getsize(path) == 5 * 2 * N * np.dtype(npdt).itemsize        # 2*N*sizeof(R) bytes per frame
    rx = RadioUserBinary(N, input_filename=path, n_frames=F, dtype=dtype)
    a = rx.receive()
    assert a.dtype == npdt and np.array_equal(a, x[:3])
    b = rx.receive()                       # wraps at EOF (auto_reset default true)
    assert np.array_equal(b[:2], x[3:5]) and np.array_equal(b[2], x[0])
    rx.close()


def test_eof_without_auto_reset_aborts(tmp_path):
    N = 4
    path = str(tmp_path / "iq.bin")
    RadioUserBinary(N, output_filename=path).send(np.arange(2 * 2 * N, dtype=np.float32))
    rx = RadioUserBinary(N, input_filename=path, auto_reset=False, n_frames=1)
    rx.receive(); rx.receive()
    with pytest.raises(ProcessingAborted):
        rx.receive()
    assert rx.is_done()
    with pytest.raises(RuntimeError, match="failbit"):
        RadioUserBinary(N, input_filename=str(tmp_path / "missing.bin"))
    with pytest.raises(RuntimeError, match="not open"):
        RadioUserBinary(N).receive()
