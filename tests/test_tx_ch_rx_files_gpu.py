"""The reference's README workflow (README.md:151-169) with the GPU work-alikes: dvbs2_tx -> out_tx.bin -> dvbs2_ch ->
out_tx_noisy.bin -> dvbs2_rx, through raw IQ files in the reference's format."""
import io
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_src_file_roundtrip(tmp_path):
    from dvbs2_amd.srcfile import SourceUser, load_src, save_src
    bits = np.unpackbits(np.load(os.path.join(GOLD, "src_K_14232.npy")))[:14232].astype(np.int32)
    p = str(tmp_path / "K_14232.src")
    save_src(p, bits)
    txt = open(p).read().split()
    assert txt[0] == "1" and txt[1] == "14232"                  # the layout of conf/src/K_14232.src
    assert np.array_equal(load_src(p, 14232)[0], bits)
    s = SourceUser(p, 14232)
    assert np.array_equal(s.generate(3), np.stack([bits] * 3))
    with pytest.raises(ValueError):
        load_src(p, 9552)


@pytest.mark.gpu
@pytest.mark.parametrize("fine", [False, True])
def test_tx_ch_rx_through_files(tmp_path, fine):
    from dvbs2_amd import ch, rx, tx
    from dvbs2_amd.srcfile import save_src
    bits = np.unpackbits(np.load(os.path.join(GOLD, "src_K_14232.npy")))[:14232].astype(np.int32)
    src = str(tmp_path / "K_14232.src")
    save_src(src, bits)
    f_tx, f_noisy, f_snk = (str(tmp_path / n) for n in ("out_tx.bin", "out_tx_noisy.bin", "sink.u8"))
    log = io.StringIO()
    n_tx = tx.run(tx.build_parser().parse_args(["--rad-type", "USER_BIN", "--rad-tx-file-path", f_tx, "-F", "8", "--src-type", "USER", "--src-path", src,
                                                "--mod-cod", "QPSK-S_8/9", "--n-frames", "24"]), out=log)
    assert n_tx == 24 and os.path.getsize(f_tx) == 24 * 2 * 8370 * 2 * 4
    # (8 dB is far above the waterfall: this test is about the files.  The channel's sigma at the sample rate IS the symbol-rate one behind the unit-energy matched filter,
    # TX_RX/main.cpp:408-409 and results/r06/filtered_loop.md)
    assert ch.run(ch.build_parser().parse_args(["--rad-rx-file-path", f_tx, "--rad-tx-file-path", f_noisy, "--rad-rx-no-loop", "-F", "8",
                                                "--mod-cod", "QPSK-S_8/9", "-m", "8.0"]), out=log) == 24
    argv = ["--src-type", "USER", "--src-path", src, "--rad-type", "USER_BIN", "--rad-rx-file-path", f_noisy, "-F", "8", "--mod-cod", "QPSK-S_8/9",
            "--dec-implem", "NMS", "--dec-ite", "10", "--dec-simd", "INTER", "--snk-path", f_snk, "--rad-rx-no-loop", "--no-wl-phases"]
    st = rx.run(rx.build_parser().parse_args(argv + (["--sync-fine"] if fine else [])), out=log)
    assert st["frames"] >= 16 and st["locked_frames"] >= st["frames"] - 6 and st["be"] == 0 and st["fe"] == 0, log.getvalue()
    assert st["delay"] == 0                                     # the extraction starts behind the two 20-symbol group delays: frames arrive aligned
    assert os.path.getsize(f_snk) == st["frames"] * 14232 // 8          # Sink_user_binary: eight payload bits per byte
    got = np.unpackbits(np.fromfile(f_snk, dtype=np.uint8), bitorder="little").reshape(-1, 14232)
    assert (got[-8:] == bits[None, :]).all()


def test_binary_source_and_sink_are_inverse_and_loop_like_the_reference(tmp_path):
    """Source_user_binary / Sink_user_binary (DVBS2.cpp:369,385; README.md:203-210, the video workflow): eight payload bits per byte; a looping source wraps around in the middle of a
    frame, a non-looping one pads its last frame and is done; what the sink writes is what the source read."""
    from dvbs2_amd.srcfile import SinkUserBinary, SourceAZCW, SourceDone, SourceUserBinary
    K = 14232
    data = np.random.default_rng(1).integers(0, 256, 5000, dtype=np.uint8)          # 2.81 frames of 1779 bytes
    p, q = str(tmp_path / "in.ts"), str(tmp_path / "out.ts")
    data.tofile(p)
    s = SourceUserBinary(p, K, auto_reset=True)
    assert s.frames_left() is None
    fr = np.concatenate([s.generate(2), s.generate(2), s.generate(3)])
    assert fr.shape == (7, K) and set(np.unique(fr)) <= {0, 1}
    assert fr[0, 0] == (data[0] & 1) and fr[0, 7] == (data[0] >> 7) and fr[0, 8] == (data[1] & 1)          # the first bit is the least significant one
    snk = SinkUserBinary(q, K); snk.send(fr[:4]); snk.send(fr[4:]); snk.close()
    out = np.fromfile(q, dtype=np.uint8)
    assert out.size == 7 * K // 8 and np.array_equal(out, np.tile(data, 3)[: out.size])                    # the file over and over, across frame borders
    once = SourceUserBinary(p, K, auto_reset=False)
    assert once.frames_left() == 3
    a = once.generate(2); assert once.frames_left() == 1
    b = once.generate(2)                                                                                  # the third frame is the file's tail + zeros, the fourth is padding
    assert np.array_equal(np.packbits(np.concatenate([a, b]).astype(np.uint8), axis=1, bitorder="little").ravel()[:5000], data) and not b[0, (5000 - 2 * 1779) * 8:].any() and not b[1].any()
    with pytest.raises(SourceDone):
        once.generate(1)
    assert not SourceAZCW(K).generate(3).any()
    with pytest.raises(ValueError):
        SourceUserBinary(p, 14231)


@pytest.mark.gpu
def test_a_file_sent_with_user_bin_comes_out_of_the_sink(tmp_path):
    """README.md:199-217 of the reference (a transport stream through dvbs2_tx --src-type USER_BIN ... dvbs2_rx --snk-path) with the GPU work-alikes and a noisy channel file in
    between: behind the frames the synchronizer needs to lock, the sink holds the file, byte for byte, over and over."""
    from dvbs2_amd import ch, rx, tx
    data = np.random.default_rng(2).integers(0, 256, 3 * 1779 + 600, dtype=np.uint8)                       # not a whole number of frames
    f_in, f_tx, f_noisy, f_snk = (str(tmp_path / n) for n in ("video.ts", "out_tx.bin", "out_tx_noisy.bin", "output.ts"))
    data.tofile(f_in)
    log = io.StringIO()
    assert tx.run(tx.build_parser().parse_args(["--rad-tx-file-path", f_tx, "-F", "8", "--src-type", "USER_BIN", "--src-path", f_in, "--mod-cod", "QPSK-S_8/9", "--n-frames", "32"]), out=log) == 32
    assert ch.run(ch.build_parser().parse_args(["--rad-rx-file-path", f_tx, "--rad-tx-file-path", f_noisy, "--rad-rx-no-loop", "-F", "8", "--mod-cod", "QPSK-S_8/9", "-m", "4.5"]), out=log) == 32
    st = rx.run(rx.build_parser().parse_args(["--src-type", "NONE", "--rad-rx-file-path", f_noisy, "-F", "8", "--mod-cod", "QPSK-S_8/9", "--dec-implem", "NMS", "--dec-ite", "10",
                                               "--snk-path", f_snk, "--rad-rx-no-loop"]), out=log)
    out = np.fromfile(f_snk, dtype=np.uint8)
    assert out.size == st["frames"] * 1779 and st["frames"] >= 24
    sent = np.tile(data, 32 * 1779 // data.size + 2)[: 32 * 1779]
    tail = out[-16 * 1779:]                                                                                # the last 16 frames the receiver put out: all behind the lock
    k = [i for i in range(0, sent.size - tail.size + 1, 1779) if np.array_equal(sent[i:i + tail.size], tail)]
    assert len(k) == 1, log.getvalue()


@pytest.mark.gpu
def test_tx_once_through_a_file_ends_by_itself(tmp_path):
    from dvbs2_amd import tx
    data = np.arange(2 * 1779 + 10, dtype=np.uint32).astype(np.uint8)
    f_in, f_tx = str(tmp_path / "in.bin"), str(tmp_path / "out_tx.bin")
    data.tofile(f_in)
    n = tx.run(tx.build_parser().parse_args(["--rad-tx-file-path", f_tx, "-F", "2", "--src-type", "USER_BIN", "--src-no-loop", "--src-path", f_in, "--mod-cod", "QPSK-S_8/9"]), out=io.StringIO())
    assert n == 4 and os.path.getsize(f_tx) == 4 * 2 * 8370 * 2 * 4                                        # three frames of payload in two calls of two
