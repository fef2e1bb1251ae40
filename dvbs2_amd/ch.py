"""`dvbs2_ch` work-alike (row N3): the AWGN channel between two raw IQ files, noise added on the GPU.

Mirrors /root/reference src/mains/CH/main.cpp:25-104 for `--chn-type AWGN` (its default): a sequence
receive -> add_noise -> send that runs until the input file ends, sigma from `-m` (Eb/N0 in dB) through
ebn0_to_esn0 / esn0_to_sigma with the code rate K_bch / N_ldpc (main.cpp:35-42), frames of
p_rad.N = pl_frame * osf complex samples (DVBS2.cpp:175).  The SYNCHRO channel (fading, Farrow fractional
delay, frequency shift: main.cpp:56-65) is sample-serial test-bench code and is not provided.

  python -m dvbs2_amd.ch --mod-cod QPSK-S_8/9 -m 4.5 --rad-rx-file-path after_TX.bin \
         --rad-tx-file-path before_RX_4.5dB.bin --rad-rx-no-loop
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import params as P
from .iqfile import ProcessingAborted, RadioUserBinary


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="dvbs2_ch", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mod-cod", default="QPSK-S_8/9")
    ap.add_argument("-m", "--sim-noise-min", type=float, default=3.2, dest="ebn0", help="Eb/N0 in dB (DVBS2.cpp: sim-noise-min)")
    ap.add_argument("-F", "--src-fra", type=int, default=1, dest="n_frames", help="frames per sequence iteration")
    ap.add_argument("--shp-osf", type=int, default=2, dest="osf", help="samples per symbol (Shaping_filter.hpp:27)")
    ap.add_argument("--rad-rx-file-path", required=True)
    ap.add_argument("--rad-tx-file-path", required=True)
    ap.add_argument("--rad-rx-no-loop", action="store_true", help="stop at the end of the input file instead of rewinding")
    ap.add_argument("--rad-type", default="USER_BIN", choices=["USER_BIN"])
    ap.add_argument("--chn-type", default="AWGN", choices=["AWGN"])
    ap.add_argument("--sim-seed", type=int, default=0, dest="seed")
    ap.add_argument("--max-frames", type=int, default=0, help="stop after this many frames (needed when the input loops)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--sim-stats", action="store_true", help="per-kernel-group device time at the end (the reference's --sim-stats)")
    return ap


def run(args, out=sys.stdout) -> int:
    """-> number of frames written"""
    from .receiver import Dvbs2Hip                       # needs the GPU library: no CPU fallback
    mc = P.get_modcod(args.mod_cod)
    N = mc.pl_frame * args.osf                           # p_rad.N, complex samples per frame
    sigma = P.esn0_to_sigma(P.ebn0_to_esn0(args.ebn0, mc.K_bch / mc.N_ldpc, mc.bps))
    rcv = RadioUserBinary(N, input_filename=args.rad_rx_file_path, auto_reset=not args.rad_rx_no_loop, n_frames=args.n_frames)
    snd = RadioUserBinary(N, output_filename=args.rad_tx_file_path, n_frames=args.n_frames)
    rx = Dvbs2Hip(mc.name, max_frames=args.n_frames, device=args.device)
    if args.sim_stats:
        rx.timing_enable(True)
    print("Channel AWGN", file=out)
    frames, call = 0, 0
    try:
        while not args.max_frames or frames < args.max_frames:
            try:
                x = rcv.receive()
            except ProcessingAborted:
                break
            y = rx.add_noise(np.float32(sigma), x.astype(np.float32, copy=False), seed=(args.seed << 32) + call, n_frames=args.n_frames)
            snd.send(y)
            frames += args.n_frames
            call += 1
    finally:
        if args.sim_stats:
            from .sim import print_stats
            print_stats([rx], out)
        rx.close(); rcv.close(); snd.close()
    mb = 2 * N * frames * 4 / (1024 * 1024)
    print("Samples size: %d MB" % mb, file=out)
    print("(II) The samples are being written in the '%s' file... " % args.rad_tx_file_path, file=out)
    return frames


if __name__ == "__main__":
    run(build_parser().parse_args())
