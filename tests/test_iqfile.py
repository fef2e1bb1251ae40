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


def test_ch_parser_matches_the_reference_command_line():
    """host logic of the `dvbs2_ch` work-alike (README.md:151-169 workflow); the noise itself needs the GPU"""
    from dvbs2_amd import ch
    a = ch.build_parser().parse_args(["--rad-type", "USER_BIN", "--rad-rx-file-path", "after_TX.bin", "--rad-tx-file-path", "before_RX_4.5dB.bin",
                                      "--rad-rx-no-loop", "-m", "4.5"])
    assert a.ebn0 == 4.5 and a.rad_rx_no_loop and a.chn_type == "AWGN" and a.osf == 2 and a.n_frames == 1
    with pytest.raises(SystemExit):
        ch.build_parser().parse_args(["--rad-rx-file-path", "x", "--rad-tx-file-path", "y", "--chn-type", "SYNCHRO"])


@pytest.mark.gpu
def test_ch_adds_awgn_of_the_requested_variance(tmp_path):
    import io
    from dvbs2_amd import ch, params as P
    mc = P.get_modcod("QPSK-S_8/9")
    N = mc.pl_frame * 2
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, 2 * N)).astype(np.float32)
    src, dst = str(tmp_path / "after_TX.bin"), str(tmp_path / "before_RX.bin")
    RadioUserBinary(N, output_filename=src).send(x)
    args = ch.build_parser().parse_args(["--rad-rx-file-path", src, "--rad-tx-file-path", dst, "--rad-rx-no-loop", "-m", "4.5", "-F", "2"])
    log = io.StringIO()
    assert ch.run(args, out=log) == 6 and "Channel AWGN" in log.getvalue()
    y = RadioUserBinary(N, input_filename=dst, n_frames=6).receive()
    sigma = P.esn0_to_sigma(P.ebn0_to_esn0(4.5, mc.K_bch / mc.N_ldpc, mc.bps))
    n = (y - x).astype(np.float64)
    assert abs(n.mean()) < 5e-3 and abs(n.std() - sigma) < 5e-3 * sigma + 2e-3
    assert not np.array_equal(n[0], n[2])                      # a fresh noise block per sequence iteration


def test_the_reference_readme_command_lines_parse():
    """README.md:156,162,168 of the reference (the file workflow) and :203,210 (the transport-stream workflow, radio flags left out: there is no USRP here), word for word behind the
    program names: the work-alikes take them."""
    from dvbs2_amd import ch, rx, tx
    a = tx.build_parser().parse_args("--sim-stats --rad-type USER_BIN --rad-tx-file-path out_tx.bin -F 8 --src-type USER --src-path ../conf/src/K_14232.src --mod-cod QPSK-S_8/9 --tx-time-limit 10000".split())
    assert a.sim_stats and a.tx_time_limit == 10000 and a.n_frames_batch == 8
    b = ch.build_parser().parse_args("--sim-stats --rad-type USER_BIN --rad-rx-file-path out_tx.bin --rad-tx-file-path out_tx_noisy.bin --rad-rx-no-loop -F 8 --mod-cod QPSK-S_8/9 -m 4.0".split())
    assert b.ebn0 == 4.0 and b.rad_rx_no_loop
    c = rx.build_parser().parse_args(("--sim-stats --src-type USER --src-path ../conf/src/K_14232.src --rad-type USER_BIN --rad-rx-file-path out_tx_noisy.bin -F 8 --mod-cod QPSK-S_8/9 "
                                      "--dec-implem NMS --dec-ite 10 --dec-simd INTER --snk-path /dev/null --rad-rx-no-loop --no-wl-phases").split())
    assert c.dec_implem == "NMS" and c.dec_ite == 10 and c.snk_path == "/dev/null"
    d = tx.build_parser().parse_args("--sim-stats -F 8 --src-type USER_BIN --src-path /path/to/input/ts/video.ts --mod-cod QPSK-S_8/9 --rad-tx-file-path out_tx.bin".split())
    assert d.src_type == "USER_BIN"
    e = rx.build_parser().parse_args("--sim-stats -F 16 --mod-cod QPSK-S_8/9 --dec-implem NMS --dec-ite 10 --dec-simd INTER --snk-path output_stream_fifo.ts --rad-rx-file-path out_tx_noisy.bin".split())
    assert e.snk_path == "output_stream_fifo.ts"


def test_the_reference_trace_command_lines_parse_in_the_simulator():
    """the `command=` lines of all eleven traces under refs/ (tests/golden/refs_tx_rx_bb.json, refs_tx_rx.json), word for word behind the program name: `python -m dvbs2_amd.sim`
    takes the dvbs2_tx_rx_bb ones as they are and the dvbs2_tx_rx ones with --perfect-sync (the sample-serial synchronizers those five traces run are out of scope: without the
    genie a channel delay is refused, before anything touches a GPU)."""
    import json, os
    from dvbs2_amd import sim
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    n = 0
    for fn, prog in (("refs_tx_rx_bb.json", "dvbs2_tx_rx_bb"), ("refs_tx_rx.json", "dvbs2_tx_rx")):
        for name, t in json.load(open(os.path.join(gold, fn))).items():
            words = t["command"].split()
            assert words[0] == prog
            a = sim.build_parser().parse_args(words[1:] + (["--perfect-sync"] if prog == "dvbs2_tx_rx" else []))
            assert a.sim_noise_min == t["rows"][0]["ebn0"] and abs(a.sim_noise_max - 0.01 - t["rows"][-1]["ebn0"]) < 1e-9, name
            n += 1
            if prog == "dvbs2_tx_rx":
                with pytest.raises(SystemExit, match="perfect-sync"):
                    sim.run(sim.build_parser().parse_args(words[1:]))
    assert n == 11
