"""`dvbs2_rx` work-alike WITHOUT the sample-serial loops (timing recovery, coarse frequency: out of scope, SURVEY.md
section 8): raw IQ file -> front gain stage (Multiplier_AGC, RX/main_sched.cpp:197) -> coarse frequency shift (--coarse-freq; :198) -> matched filter (a5) -> extraction at a known symbol phase -> gain stage
(main_sched.cpp:205) -> frame synchronizer (N4) -> pilot-aided phase synchronizer (N4, optional) -> fused RX chain (a7 .. a8) -> monitor against the source pattern -> sink.  It serves
files made by `dvbs2_amd.tx` / `dvbs2_amd.ch` (or by the reference's dvbs2_tx / dvbs2_ch without timing or frequency
offsets): README.md:151-169 of the reference.

  python -m dvbs2_amd.rx --src-type USER --src-path K_14232.src --rad-rx-file-path out_tx_noisy.bin -F 8 \
         --mod-cod QPSK-S_8/9 --dec-implem NMS --dec-ite 10 --snk-path /dev/null --rad-rx-no-loop
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import params as P
from .iqfile import ProcessingAborted, RadioUserBinary
from .srcfile import SinkUserBinary, load_src


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="dvbs2_rx", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mod-cod", default="QPSK-S_8/9")
    ap.add_argument("-F", "--src-fra", type=int, default=1, dest="n_frames_batch")
    ap.add_argument("--src-type", default="USER", choices=["USER", "NONE"], help="USER: count errors against the pattern file")
    ap.add_argument("--src-path", default="")
    ap.add_argument("--shp-osf", type=int, default=2, dest="osf")
    ap.add_argument("--rad-type", default="USER_BIN", choices=["USER_BIN"])
    ap.add_argument("--rad-rx-file-path", required=True)
    ap.add_argument("--rad-rx-no-loop", action="store_true")
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--dec-implem", default="SPA", choices=["NMS", "MS", "SPA", "SPA_TANH", "SPA_EXACT"])    # the reference's defaults (DVBS2.cpp:135-138)
    ap.add_argument("--dec-ite", type=int, default=50)
    ap.add_argument("--dec-alpha", type=float, default=1.0)
    ap.add_argument("--dec-simd", default="", help="accepted and ignored (the GPU batches frames with -F)")
    ap.add_argument("--no-wl-phases", action="store_true", help="accepted: there are no waiting / learning phases here")
    ap.add_argument("--snk-path", default="", help="decoded payload of every frame, eight bits per byte (the reference's Sink_user_binary: a file sent with dvbs2_tx --src-type USER_BIN comes out as it went in)")
    ap.add_argument("--timing-offset", type=int, default=-1, help="sample index of the first symbol after the matched filter (default: two group delays)")
    ap.add_argument("--sync-fine", action="store_true", help="run the pilot-aided phase synchronizer before the chain")
    ap.add_argument("--coarse-freq", type=float, default=0.0, help="carrier offset of the received samples in cycles per sample: the coarse frequency synchronizer's task of the transmission "
                                                                   "phase (the frequency shift) with this as its loop's frozen estimate; the loop itself is sample-serial and out of scope")
    ap.add_argument("--no-agc", action="store_true", help="leave out the two gain stages of the reference's graph (front_agc on the samples, mult_agc on the symbols)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--sim-stats", action="store_true", help="per-kernel-group device time at the end (the reference's --sim-stats)")
    return ap


def run(args, out=sys.stdout) -> dict:
    from .receiver import Dvbs2Hip
    mc = P.get_modcod(args.mod_cod)
    F, n, osf = args.n_frames_batch, mc.pl_frame, args.osf
    if not args.rad_rx_no_loop and not args.max_frames:
        raise ValueError("a looping input needs --max-frames")
    pattern = load_src(args.src_path, mc.K_bch) if args.src_type == "USER" and args.src_path else None
    rx = Dvbs2Hip(mc.name, max_frames=F, n_ite=args.dec_ite, alpha=args.dec_alpha, early_stop=True, implem=args.dec_implem, device=args.device)
    if args.sim_stats:
        rx.timing_enable(True)
    if args.coarse_freq:
        rx.sync_coarse_set_freq(args.coarse_freq)
    rcv = RadioUserBinary(n * osf, input_filename=args.rad_rx_file_path, auto_reset=not args.rad_rx_no_loop, n_frames=F)
    snk = SinkUserBinary(args.snk_path, mc.K_bch) if args.snk_path else None
    off = args.timing_offset if args.timing_offset >= 0 else 2 * 20 * osf          # two group delays of grp_delay * osf samples
    tail = np.zeros((0, 2), np.float32)                                            # matched-filter samples not yet turned into symbols
    skip = off
    st = dict(frames=0, locked_frames=0, be=0, fe=0, delay=None)
    stable = 0                                                                     # frames since the synchronizer's delay last moved
    try:
        while not args.max_frames or st["frames"] < args.max_frames:
            try:
                x = rcv.receive()
            except ProcessingAborted:
                break
            x = x.astype(np.float32, copy=False)
            if not args.no_agc:
                x = rx.agc(x, n_frames=F, output_energy=1.0 / osf)                  # front_agc: DVBS2.cpp:660-664
            if args.coarse_freq:
                _, _, x = rx.sync_coarse_synchronize(x, n_frames=F)                  # sync_coarse_f: RX/main_sched.cpp:198-200
            mf = np.concatenate([tail, rx.filter(x, n_frames=F).reshape(-1, 2)])
            mf, skip = mf[skip:], 0                                                # perfect timing: every osf-th sample from `off`
            n_sym = (mf.shape[0] // osf // n) * n                                  # whole frames of symbols
            if n_sym == 0:
                tail = mf
                continue
            sym, tail = mf[: n_sym * osf : osf], mf[n_sym * osf:]
            for b0 in range(0, n_sym // n, F):
                blk = np.ascontiguousarray(sym[b0 * n:(b0 + F) * n])
                Fb = blk.shape[0] // n
                if not args.no_agc:
                    blk = rx.agc(blk, n_frames=Fb, output_energy=1.0)                   # mult_agc: DVBS2.cpp:653-657
                delay, flags, tri, aligned = rx.sync_frame_synchronize(blk.reshape(Fb, 2 * n), with_flags=True)
                if args.sync_fine:
                    # the reference's task order (src/mains/RX/main.cpp): PL descramble -> fine synchronizer -> remove PLH ->
                    # estimate -> demodulate + deinterleave -> LDPC -> BCH -> BB descramble, one C-ABI call per task
                    _, _, fixed = rx.sync_freq_phase_synchronize(rx.pl_descramble(aligned))
                    xf = rx.remove_plh(fixed)
                    sig, _, _ = rx.estimate(xf)
                    vk, _ = rx.decode_siho(rx.demodulate(sig, xf, deinterleave=True))
                    bits = rx.bb_descramble(rx.decode_hiho(vk)[0])
                else:
                    bits, _, _ = rx.rx_bb(aligned)
                for f in range(Fb):
                    st["frames"] += 1
                    stable = stable + 1 if st["delay"] is not None and delay[f] == st["delay"] else 0
                    locked = stable >= 2                                           # the delay line has settled on this alignment
                    st["delay"] = int(delay[f])
                    if snk:
                        snk.send(bits[f])
                    if pattern is not None and locked:
                        e = min(int((bits[f] != p).sum()) for p in pattern)
                        st["locked_frames"] += 1; st["be"] += e; st["fe"] += e > 0
    finally:
        if args.sim_stats:
            from .sim import print_stats
            print_stats([rx], out)
        rx.close(); rcv.close()
        if snk:
            snk.close()
    st["ber"] = st["be"] / max(1, st["locked_frames"] * mc.K_bch)
    st["fer"] = st["fe"] / max(1, st["locked_frames"])
    print("# frames %(frames)d | in lock %(locked_frames)d | BE %(be)d | FE %(fe)d | BER %(ber).2e | FER %(fer).2e | delay %(delay)s" % st, file=out)
    return st


if __name__ == "__main__":
    run(build_parser().parse_args())
