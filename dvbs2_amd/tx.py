"""`dvbs2_tx` work-alike (rows N1 + N2 + N3): source -> BB scramble -> BCH -> LDPC -> interleave -> modulate -> frame ->
PL scramble -> shaping filter -> raw IQ file, every stage on the GPU (src/mains/TX/main.cpp of the reference; README.md:151-158).

  python -m dvbs2_amd.tx --rad-tx-file-path out_tx.bin -F 8 --src-type USER --src-path K_14232.src --mod-cod QPSK-S_8/9 --n-frames 64
"""
from __future__ import annotations

import argparse
import sys
import time

from . import params as P
from .iqfile import RadioUserBinary
from .srcfile import SourceAZCW, SourceDone, SourceUser, SourceUserBinary


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="dvbs2_tx", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mod-cod", default="QPSK-S_8/9")
    ap.add_argument("-F", "--src-fra", type=int, default=1, dest="n_frames_batch")
    ap.add_argument("--src-type", default="RAND", choices=["RAND", "USER", "USER_BIN", "AZCW"])      # DVBS2.cpp:66
    ap.add_argument("--src-path", default="")
    ap.add_argument("--src-no-loop", action="store_true", help="USER_BIN: send the file once (the last frame zero-padded) instead of over and over")
    ap.add_argument("--shp-osf", type=int, default=2, dest="osf", choices=[2], help="samples per symbol (the GPU shaping filter is built for 2)")
    ap.add_argument("--rad-type", default="USER_BIN", choices=["USER_BIN"])
    ap.add_argument("--rad-tx-file-path", required=True)
    ap.add_argument("--n-frames", type=int, default=0, help="stop after this many frames")
    ap.add_argument("--tx-time-limit", type=float, default=0.0, help="stop after this many milliseconds")
    ap.add_argument("--sim-seed", type=int, default=0, dest="seed")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--sim-stats", action="store_true", help="per-kernel-group device time at the end (the reference's --sim-stats)")
    return ap


def run(args, out=sys.stdout) -> int:
    """-> number of frames written"""
    from .receiver import Dvbs2Hip
    mc = P.get_modcod(args.mod_cod)
    once = args.src_type == "USER_BIN" and args.src_no_loop
    if not args.n_frames and not args.tx_time_limit and not once:
        raise ValueError("one of --n-frames / --tx-time-limit is needed to end the transmission")
    if args.src_type in ("USER", "USER_BIN") and not args.src_path:
        raise ValueError("--src-type %s needs --src-path" % args.src_type)
    F = args.n_frames_batch
    src = (SourceUser(args.src_path, mc.K_bch) if args.src_type == "USER" else SourceUserBinary(args.src_path, mc.K_bch, auto_reset=not args.src_no_loop) if args.src_type == "USER_BIN"
           else SourceAZCW(mc.K_bch) if args.src_type == "AZCW" else None)
    rx = Dvbs2Hip(mc.name, max_frames=F, device=args.device)
    if args.sim_stats:
        rx.timing_enable(True)
    snd = RadioUserBinary(mc.pl_frame * args.osf, output_filename=args.rad_tx_file_path, n_frames=F)
    t0, frames, call = time.perf_counter(), 0, 0
    try:
        while (not args.n_frames or frames < args.n_frames) and (not args.tx_time_limit or (time.perf_counter() - t0) * 1e3 < args.tx_time_limit):
            try:
                info = src.generate(F) if src else None
            except SourceDone:
                break
            _, pl = rx.tx_bb(F, info=info, seed=(args.seed << 32) + call)
            snd.send(rx.shape_filter(pl, n_frames=F, osf=args.osf))        # the filter memory carries over from call to call
            frames += F
            call += 1
    finally:
        if args.sim_stats:
            from .sim import print_stats
            print_stats([rx], out)
        rx.close(); snd.close()
    print("(II) %d frames of %d complex samples written in '%s'" % (frames, mc.pl_frame * args.osf, args.rad_tx_file_path), file=out)
    return frames


if __name__ == "__main__":
    run(build_parser().parse_args())
