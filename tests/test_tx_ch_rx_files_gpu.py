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
    # noise of the requested Eb/N0 per SYMBOL needs sqrt(osf) more per sample ahead of a unit-gain matched filter: 4.0 dB + 3 dB
    assert ch.run(ch.build_parser().parse_args(["--rad-rx-file-path", f_tx, "--rad-tx-file-path", f_noisy, "--rad-rx-no-loop", "-F", "8",
                                                "--mod-cod", "QPSK-S_8/9", "-m", "8.0"]), out=log) == 24
    argv = ["--src-type", "USER", "--src-path", src, "--rad-type", "USER_BIN", "--rad-rx-file-path", f_noisy, "-F", "8", "--mod-cod", "QPSK-S_8/9",
            "--dec-implem", "NMS", "--dec-ite", "10", "--dec-simd", "INTER", "--snk-path", f_snk, "--rad-rx-no-loop", "--no-wl-phases"]
    st = rx.run(rx.build_parser().parse_args(argv + (["--sync-fine"] if fine else [])), out=log)
    assert st["frames"] >= 16 and st["locked_frames"] >= st["frames"] - 6 and st["be"] == 0 and st["fe"] == 0, log.getvalue()
    assert st["delay"] == 0                                     # the extraction starts behind the two 20-symbol group delays: frames arrive aligned
    assert os.path.getsize(f_snk) == st["frames"] * 14232
