#!/usr/bin/env python3
"""FER of the chain with the IN-SCOPE synchronizers in the loop (frame synchronizer, Luise-Reggiannini fine frequency, pilot-aided phase), beside the
genie-timed loop of results/r06/filtered_loop.md -- what the block-wise synchronizers cost at the operating point, the reference's sample-serial loops (timing, coarse
frequency, AGC) being replaced by a genie.  GPU box:

    python tools/sync_in_loop.py --ebn0 3.7 3.8 --fe 400 --max-frames 200000 --json gpurun_out/sync_in_loop.json

One continuous stream per noise point, as a receiver sees it: the fixed payload conf/src/K_14232.src in every frame (what the reference's dvbs2_rx counts errors against,
RX/main.cpp: Source_user), an unknown frame start (--off symbols), a constant carrier phase and a residual frequency offset (--freq cycles per symbol: what a coarse loop
leaves behind) -> shaping filter -> AWGN at the sample rate -> matched filter -> every second sample (timing by genie) -> the tasks of the reference's RX graph in its order,
one C-ABI call per task: frame synchronizer -> PL descrambler -> L&R -> pilot-aided phase -> remove PLH -> estimate -> demodulate + de-interleave -> LDPC -> BCH -> BB
descrambler.  Variants: `frame` (frame synchronizer only, no rotation applied: its cost alone), `fine` (rotation applied; L&R + phase synchronizer correct it).
Filters, synchronizers and the delay line keep their state from call to call, so the stream is continuous across the calls."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_point(Rx, P, mc, ebn0, variant, a):
    F, n, off = a.F, mc.pl_frame, a.off
    rx = Rx(mc.name, max_frames=F, n_ite=50, alpha=1.0, early_stop=True, implem="SPA")
    pattern = np.unpackbits(np.load(os.path.join(ROOT, "tests", "golden", "src_K_%d.npy" % (14232 if mc.K_bch == 14232 else 9552))))[:mc.K_bch].astype(np.int32)      # DVBS2.cpp:336-349
    _, pl = rx.tx_bb(1, info=pattern[None, :])
    frame = pl.reshape(n, 2).astype(np.float64)
    frame = frame[:, 0] + 1j * frame[:, 1]
    sigma = np.float32(P.esn0_to_sigma(P.ebn0_to_esn0(ebn0, mc.code_rate, mc.bps)))
    rot = variant == "fine"
    st = dict(frames=0, counted=0, be=0, fe=0, delay=None, stable=0, moved=0)
    t0, k = time.time(), 0
    idx = (np.arange(F * n) - off) % n
    while st["fe"] < a.fe and st["counted"] < a.max_frames:
        t = np.arange(F * n, dtype=np.float64) + float(k) * F * n                      # absolute symbol index of this call's stretch of the stream
        s = frame[idx]                                                                 # F n is a multiple of n: the same indices every call
        if rot:
            s = s * np.exp(1j * (a.phase + 2.0 * np.pi * a.freq * t))
        x = np.empty((F * n, 2), np.float32); x[:, 0] = s.real; x[:, 1] = s.imag
        up = rx.shape_filter(x, n_frames=F, osf=2)
        noisy = rx.add_noise(sigma, up, seed=(a.seed << 20) + k, n_frames=F)
        if a.agc:
            noisy = rx.agc(noisy, n_frames=F, output_energy=0.5)                        # front_agc (RX/main_sched.cpp:197; DVBS2.cpp:660-664)
        mf = rx.filter(noisy, n_frames=F).reshape(-1, 2)
        sym = np.ascontiguousarray(mf[0::2]).reshape(F, 2 * n)                         # the two filters delay the stream by 40 symbols: part of the unknown frame start
        if a.agc:
            sym = rx.agc(sym, n_frames=F, output_energy=1.0).reshape(F, 2 * n)          # mult_agc (main_sched.cpp:205; DVBS2.cpp:653-657)
        delay, flags, tri, aligned = rx.sync_frame_synchronize(sym, with_flags=True)
        if variant == "frame":
            bits, _, _ = rx.rx_bb(aligned, sigma=sigma if a.est_perfect else None)
        else:
            desc = rx.pl_descramble(aligned)
            _, _, desc = rx.sync_lr_synchronize(desc)
            _, _, fixed = rx.sync_freq_phase_synchronize(desc)
            xf = rx.remove_plh(fixed)
            sg = np.full(F, sigma, np.float32) if a.est_perfect else rx.estimate(xf)[0]
            vk, _ = rx.decode_siho(rx.demodulate(sg, xf, deinterleave=True))
            bits = rx.bb_descramble(rx.decode_hiho(vk)[0])
        err = (bits != pattern[None, :]).sum(axis=1)
        for f in range(F):
            st["frames"] += 1
            same = st["delay"] is not None and delay[f] == st["delay"]
            st["stable"] = st["stable"] + 1 if same else 0
            if not same and st["frames"] > 8:
                st["moved"] += 1                                                        # the synchronizer left its alignment after the acquisition
            st["delay"] = int(delay[f])
            if st["frames"] > a.skip:                                                   # every frame after the acquisition counts, locked or not (a lost lock is a lost frame)
                st["counted"] += 1; st["be"] += int(err[f]); st["fe"] += int(err[f] > 0)
        k += 1
    rx.close()
    st.update(ebn0=ebn0, variant=variant, fer=st["fe"] / max(1, st["counted"]), ber=st["be"] / max(1, st["counted"] * mc.K_bch), seconds=time.time() - t0)
    return st


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--mod-cod", default="QPSK-S_8/9")
    ap.add_argument("--ebn0", type=float, nargs="+", default=[3.7, 3.8])
    ap.add_argument("--variants", nargs="+", default=["frame", "fine"])
    ap.add_argument("-F", type=int, default=256)
    ap.add_argument("--fe", type=int, default=400)
    ap.add_argument("--max-frames", type=int, default=200000)
    ap.add_argument("--skip", type=int, default=16, help="frames of acquisition at the head of the stream that are not counted")
    ap.add_argument("--off", type=int, default=1234)
    ap.add_argument("--phase", type=float, default=0.7)
    ap.add_argument("--freq", type=float, default=2e-5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--est-perfect", action="store_true", help="the channel's sigma instead of the M2M4 estimate (the reference's 16APSK trace: --est-type PERFECT)")
    ap.add_argument("--agc", action="store_true", help="the reference's two gain stages in the loop (front_agc on the samples, mult_agc on the symbols)")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from dvbs2_amd.receiver import Dvbs2Hip
    from dvbs2_amd import params as P
    mc = P.get_modcod(a.mod_cod)
    rows = []
    for e in a.ebn0:
        for v in a.variants:
            r = run_point(Dvbs2Hip, P, mc, e, v, a)
            rows.append(r)
            print("%s %.2f dB %-5s: frames %d counted %d FE %d FER %.3e BER %.2e | delay %s, moved %d times after acquisition | %.0f s" % (
                mc.name, e, v, r["frames"], r["counted"], r["fe"], r["fer"], r["ber"], r["delay"], r["moved"], r["seconds"]), flush=True)
    if a.json:
        json.dump(dict(args=vars(a), rows=rows), open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
