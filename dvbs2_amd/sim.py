"""dvbs2_tx_rx_bb work-alike: the Monte-Carlo BER/FER loop of
/root/reference src/mains/TX_RX_BB/main.cpp:139-167 with TX, channel, RX and monitor all on the
GPU(s).  Same flags (DVBS2.cpp:117-149) where they apply to this path, same table as refs/.

  python -m dvbs2_amd.sim --mod-cod QPSK-S_8/9 -m 3.6 -M 3.81 -s 0.1 --dec-implem NMS --dec-ite 10 -F 512
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m dvbs2_amd.sim --mod-cod 16APSK-N_8/9 ...

Frames shard over ranks (independent Philox streams: seed = base + rank); the monitor counters
{FRA, BE, FE} are summed over ranks once per batch (the RCCL all-reduce that replaces
tools::Monitor_reduction); every rank stops at the same batch.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np


def build_parser():
    ap = argparse.ArgumentParser(prog="dvbs2_tx_rx_bb (HIP)")
    ap.add_argument("--mod-cod", default="QPSK-S_8/9")
    ap.add_argument("-m", "--sim-noise-min", type=float, default=3.2)
    ap.add_argument("-M", "--sim-noise-max", type=float, default=6.0)
    ap.add_argument("-s", "--sim-noise-step", type=float, default=0.1)
    ap.add_argument("-e", "--max-fe", type=int, default=100)
    ap.add_argument("-F", "--sim-inter-fra", type=int, default=512, help="frames per batch per GPU (grid width)")
    ap.add_argument("--dec-ite", type=int, default=50)
    ap.add_argument("--dec-implem", default="SPA", choices=["NMS", "MS", "SPA", "SPA_TANH", "SPA_EXACT"])    # the reference's defaults: SPA, 50 ite (DVBS2.cpp:135-138)
    ap.add_argument("--dec-sched", default="QC", choices=["QC", "NATURAL"], help="NATURAL: the reference's sweep order over the rows of H (k_ldpc_nat.hip; a validation mode that wants -F 32768)")
    ap.add_argument("--dec-alpha", type=float, default=1.0)
    ap.add_argument("--no-early-stop", action="store_true")
    ap.add_argument("--est-type", default="DVBS2", choices=["DVBS2", "PERFECT"])
    ap.add_argument("--max-frames", type=int, default=10_000_000, help="cap on frames per noise point (all ranks)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--clones", type=int, default=3,
                    help="clones of the chain per process, each with its own handle, stream and -F frames in flight (the reference's Sequence runs n_threads of them, "
                         "TX_RX_BB/main.cpp:19,96; host/dvbs2_tx_rx_bb --clones): batches are dealt to them in turn and a clone's counters are read when its turn comes again, "
                         "so the next clone's kernels fill the CUs that a batch's last frames leave idle under the early stop.  1 = one batch at a time (the loop every committed sweep used)")
    ap.add_argument("--filtered", action="store_true",
                    help="TX shaping filter -> AWGN at the sample rate -> matched filter -> perfect-timing extraction "
                         "(the filtered loop of src/mains/TX_RX/main.cpp with --perfect-sync); the last frame of every batch "
                         "is cut by the filters' delay and not counted")
    ap.add_argument("--src-type", default="RAND", choices=["RAND", "USER", "USER_BIN", "AZCW"],
                    help="payload source (DVBS2.cpp:66,359-376): RAND = the TX mirror's own generator on the device; USER = a pattern file (conf/src/*.src), USER_BIN = any file, eight bits per byte, "
                         "AZCW = all-zero payloads: generated on the host and handed to the TX mirror's `info_in` socket")
    ap.add_argument("--src-path", default="")
    ap.add_argument("--src-no-loop", action="store_true", help="USER_BIN: stop a noise point when the file has been sent once")
    ap.add_argument("--ter-freq", type=int, default=500, help="accepted and ignored (a row is printed when its noise point is done)")
    ap.add_argument("--perfect-sync", action="store_true", help="dvbs2_tx_rx --perfect-sync (DVBS2.cpp:97, TX_RX/main.cpp:440): the same loop as --filtered")
    ap.add_argument("--chn-max-freq-shift", type=float, default=0.0, help="dvbs2_tx_rx's channel frequency shift: with --perfect-sync the genie removes it, so it is accepted and has no effect; "
                                                                          "without, the reference's sample-serial synchronizers would have to (out of scope, SURVEY.md 8e): refused")
    ap.add_argument("--chn-max-delay", type=float, default=0.0, help="dvbs2_tx_rx's channel delay in samples: as --chn-max-freq-shift")
    ap.add_argument("--json", default=None, help="also write the rows as JSON")
    ap.add_argument("--sim-stats", action="store_true", help="per-kernel-group device time at the end (the reference's --sim-stats, TX_RX_BB/main.cpp:110,170-178); the timers' events cost a few percent")
    return ap


def print_stats(handles, out=sys.stdout):
    """the table of --sim-stats: device time per kernel group, summed over the handles (clones) of this process"""
    tot = {}
    for h in handles:
        for name, (ms, n) in h.timing_stats().items():
            a = tot.setdefault(name, [0.0, 0])
            a[0] += ms; a[1] += n
    all_ms = sum(v[0] for v in tot.values()) or 1.0
    print("# -------------------------------------------------||------------||------------||---------", file=out)
    print("#                                     Kernel group ||   launches || device (ms) ||    share", file=out)
    print("# -------------------------------------------------||------------||------------||---------", file=out)
    for name, (ms, n) in tot.items():
        if n:
            print("# %48s || %10d || %10.2f || %6.1f %%" % (name, n, ms, 100.0 * ms / all_ms), file=out)


def run(args, out=sys.stdout):
    args.filtered = args.filtered or args.perfect_sync
    if (args.chn_max_freq_shift or args.chn_max_delay) and not args.filtered:
        raise SystemExit("a channel delay / frequency shift needs the reference's timing and coarse-frequency loops (sample-serial: out of scope) or --perfect-sync")
    import torch
    import torch.distributed as dist
    from . import params as P
    from .parallel import reduce_counters, reduce_max
    from .receiver import Dvbs2Hip
    from .srcfile import SourceDone

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("dvbs2_amd.sim needs a GPU: libdvbs2hip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    mc = P.get_modcod(args.mod_cod)
    F = args.sim_inter_fra
    alpha = 1.0 if args.dec_implem == "MS" else args.dec_alpha
    if not 1 <= args.clones <= 8:
        raise SystemExit("--clones has to be 1 .. 8")
    osf, delay = 2, 80                       # Shaping_filter.hpp:24-28: osf 2, two group delays of 20 symbols
    if args.src_type in ("USER", "USER_BIN") and not args.src_path:
        raise SystemExit("--src-type %s needs --src-path" % args.src_type)

    def make_source():
        from .srcfile import SourceAZCW, SourceUser, SourceUserBinary
        return (SourceUser(args.src_path, mc.K_bch) if args.src_type == "USER" else SourceUserBinary(args.src_path, mc.K_bch, auto_reset=not args.src_no_loop) if args.src_type == "USER_BIN"
                else SourceAZCW(mc.K_bch) if args.src_type == "AZCW" else None)
    src_box = [make_source()]                 # one source per process: its frames go to the clones in the order the batches are issued; it starts over at every noise point

    class Clone:
        def __init__(self):
            self.rx = Dvbs2Hip(mc.name, max_frames=F, n_ite=args.dec_ite, alpha=alpha, early_stop=not args.no_early_stop, device=local_rank, implem=args.dec_implem)
            if args.dec_sched == "NATURAL":
                from dvbs2_amd import lib_binding as B
                self.rx.set_ldpc_schedule(B.SCHED_NATURAL)
            self.pl = torch.empty((F, 2 * self.rx.pl_frame), dtype=torch.float32, device=dev)
            self.sent = torch.empty((F, self.rx.K_bch), dtype=torch.int32, device=dev)
            self.got = torch.empty((F, self.rx.K_bch), dtype=torch.int32, device=dev)
            self.sig = torch.empty((F,), dtype=torch.float32, device=dev)
            if args.filtered:
                self.up = torch.empty((F, 2 * self.rx.pl_frame * osf), dtype=torch.float32, device=dev)
                self.up2 = torch.empty_like(self.up)
            self.count, self.busy = [0, 0, 0], False

        def issue(self, seed):
            rx = self.rx
            info = None
            if src_box[0] is not None:
                self.info = torch.from_numpy(src_box[0].generate(F)).to(dev)          # (kept until the next batch of this clone: the copy is asynchronous to the handle's stream)
                torch.cuda.current_stream().synchronize()
                info = self.info.data_ptr()
            if args.filtered:
                rx.tx_bb_dev(info, seed, None, self.sent.data_ptr(), self.pl.data_ptr(), F)
                rx.shape_filter_dev(self.pl.data_ptr(), self.up.data_ptr(), rx.pl_frame, F)
                rx.add_noise_dev(self.sig.data_ptr(), self.up.data_ptr(), self.up2.data_ptr(), seed, 2 * rx.pl_frame * osf, F)
                rx.filter_dev(self.up2.data_ptr(), self.up.data_ptr(), rx.pl_frame * osf, F)
                rx.extract_dev(self.up.data_ptr(), self.pl.data_ptr(), rx.pl_frame, osf, delay, F)
            else:
                rx.tx_bb_dev(info, seed, self.sig.data_ptr(), self.sent.data_ptr(), self.pl.data_ptr(), F)
            rx.rx_bb_dev(self.pl.data_ptr(), self.sig.data_ptr() if args.est_type == "PERFECT" else None, self.got.data_ptr(), None, None, F)
            rx.check_errors_dev(self.sent.data_ptr(), self.got.data_ptr(), F - 1 if args.filtered else F)
            self.busy = True

    clones = [Clone() for _ in range(args.clones)]
    rx = clones[0].rx
    if args.sim_stats:
        for c in clones:
            c.rx.timing_enable(True)
    torch.cuda.synchronize()

    rows = []
    if rank == 0:
        print("# * DVB-S2 (HIP, %d GPU(s)) ---------------------------" % world, file=out)
        print("#    ** Modulation and coding = %s" % mc.name, file=out)
        print("#    ** LDPC implem           = %s (alpha %.3f, %s schedule)" % (args.dec_implem, alpha, "QC-layer" if args.dec_sched == "QC" else "natural row order"), file=out)
        print("#    ** LDPC n iterations     = %d" % args.dec_ite, file=out)
        print("#    ** Estimator             = %s" % args.est_type, file=out)
        print("#    ** Frames per batch      = %d x %d" % (F, world), file=out)
        if args.clones > 1:
            print("#    ** Clones per process     = %d" % args.clones, file=out)
        print("# ----------|----------||----------|----------|----------|----------|----------||----------|----------", file=out)
        print("#     Es/N0 |    Eb/N0 ||      FRA |       BE |       FE |      BER |      FER ||  SIM_THR |    ET/RT", file=out)
        print("#      (dB) |     (dB) ||          |          |          |          |          ||   (Mb/s) | (hhmmss)", file=out)
        print("# ----------|----------||----------|----------|----------|----------|----------||----------|----------", file=out)
    ebn0 = args.sim_noise_min
    batch_id = 0
    while ebn0 < args.sim_noise_max - 1e-9:
        esn0 = P.ebn0_to_esn0(ebn0, mc.code_rate, mc.bps)          # main.cpp:142-146
        sigma = P.esn0_to_sigma(esn0)
        src_box[0] = make_source()
        for c in clones:
            c.sig.fill_(sigma)
            c.rx.monitor_reset()
            c.count, c.busy = [0, 0, 0], False
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tot = [0, 0, 0]

        def collect(c):
            # a clone's counters run on the device from the reset above; the stopping rule sees the sum over the clones of what each last reported, reduced over the ranks
            c.count, c.busy = list(c.rx.monitor_get()), False       # syncs that clone's stream only
            return reduce_counters([sum(k.count[i] for k in clones) for i in range(3)], dev)      # 24-byte all-reduce

        turn = 0
        while True:
            c = clones[turn % len(clones)]
            turn += 1
            if c.busy:
                tot = collect(c)
            if tot[2] >= args.max_fe or tot[0] >= args.max_frames:
                break
            try:
                c.issue((args.seed << 40) + (batch_id << 8) + rank)
            except SourceDone:                                       # --src-type USER_BIN --src-no-loop: the file has been sent
                break
            batch_id += 1
        for c in clones:                                             # the batches still in flight count (the reference's threads finish theirs)
            if c.busy:
                tot = collect(c)
        et = reduce_max(time.perf_counter() - t0, dev)
        fra, be, fe = tot
        row = dict(esn0=esn0, ebn0=ebn0, fra=fra, be=be, fe=fe, ber=be / max(1, fra * mc.K_bch), fer=fe / max(1, fra),
                   thr_mbps=fra * mc.K_bch / et / 1e6, et=et)
        rows.append(row)
        if rank == 0:
            h, m_, s_ = int(et // 3600), int(et % 3600 // 60), int(et % 60)
            print("  %9.2f | %8.2f || %8d | %8d | %8d | %8.2e | %8.2e || %8.3f | %02dh%02d'%02d" % (
                esn0, ebn0, fra, be, fe, row["ber"], row["fer"], row["thr_mbps"], h, m_, s_), file=out, flush=True)
        ebn0 += args.sim_noise_step
    if rank == 0:
        print("# End of the simulation", file=out)
        if args.sim_stats:
            print_stats([c.rx for c in clones], out)
        if args.json:
            with open(args.json, "w") as fh:
                json.dump(dict(args=vars(args), n_gpus=world, rows=rows), fh, indent=1)
    for c in clones:
        c.rx.close()
    return rows


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
