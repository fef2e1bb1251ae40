#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container (it reads
/root/reference, which does not exist on the GPU box; the fixtures travel instead).

  pl_rand_seq_2bit.npy   PL scrambling sequence = the 66420 values of PL_RAND_SEQ in the reference
                         header src/common/Module/Scrambler/Scrambler_PL/Scrambler_PL.hpp:54-4207,
                         packed 4 values per byte (DATA: expected output of the Gold generator)
  refs_tx_rx_bb.json     the result rows + size headers of refs/TX_RX_BB/*.txt (the reference's
                         only regression oracle: BER/FER traces, SPA 50 ite)
  refs_tx_rx.json        the result rows of refs/TX_RX/*.txt: the reference's FULL chain (shaping filter, channel with delay / frequency shift, its sample-serial
                         synchronizers, SPA 50 ite).  The synchronizers' loops are out of scope, so these rows bound the genie-timed filtered loop
                         from above (`python make_golden.py refs_full` remakes only this one)
  src_K_14232.npy, src_K_9552.npy   conf/src/K_14232.src, K_9552.src fixed payloads (data files), packed bits (`python make_golden.py src`)
  kat_*.npz              known-answer vectors produced by the CPU oracle (seeded), small frames
                         (`python make_golden.py sync` remakes only kat_sync_frame_32apsk.npz)
  kat_ldpc_normal_8_9.npz, kat_chain_16apsk_normal_20ite.npz
                         the headline code at size (N = 64800, rate 8/9): two frames at 3.0 / 4.2 dB, NMS 10 iterations, alpha 1.0, QC and
                         NATURAL order (hard bits, CWD, iteration counts, posteriors as fp32), and BASELINE configs[3]'s chain (16APSK, NMS 20
                         iterations) for one PL frame -- so the headline parity of the GPU tests does not hang on rebuilding the oracle on the
                         GPU box (`python make_golden.py normal` remakes only these two)
"""
import json, os, re, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

def pl_seq():
    txt = open(os.path.join(REF, "src/common/Module/Scrambler/Scrambler_PL/Scrambler_PL.hpp")).read()
    i0 = txt.index("PL_RAND_SEQ")
    body = txt[txt.index("{", i0) + 1: txt.index("}", i0)]
    v = np.array([int(x) for x in re.findall(r"\d+", body)], dtype=np.uint8)
    assert v.size == 66420 and v.max() == 3
    packed = (v[0::4] | (v[1::4] << 2) | (v[2::4] << 4) | (v[3::4] << 6)).astype(np.uint8)
    np.save(os.path.join(HERE, "pl_rand_seq_2bit.npy"), packed)

def refs():
    out = {}
    for name in sorted(os.listdir(os.path.join(REF, "refs/TX_RX_BB"))):
        txt = open(os.path.join(REF, "refs/TX_RX_BB", name)).read()
        cmd = re.search(r"command=(.*)", txt).group(1).strip()
        rows = []
        for line in txt.splitlines():
            if line.startswith("#") or "|" not in line:
                continue
            f = [x.strip() for x in line.replace("||", "|").split("|")]
            rows.append(dict(esn0=float(f[0]), ebn0=float(f[1]), fra=int(f[2]), be=int(f[3]), fe=int(f[4]),
                             ber=float(f[5]), fer=float(f[6]), thr_mbps=float(f[7])))
        hdr = {}
        for key, pat in (("modcod", r"Modulation and coding = (\S+)"), ("n_cw", r"N\. cw\s+\(N\)\s+= (\d+)"),
                         ("implem", r"LDPC implem\s+= (\S+)"), ("n_ite", r"LDPC n iterations\s+= (\d+)")):
            m = re.search(pat, txt)
            hdr[key] = m.group(1) if m else None
        out[name] = dict(command=cmd, header=hdr, rows=rows)
    json.dump(out, open(os.path.join(HERE, "refs_tx_rx_bb.json"), "w"), indent=1)

def refs_full():
    out = {}
    for name in sorted(os.listdir(os.path.join(REF, "refs/TX_RX"))):
        txt = open(os.path.join(REF, "refs/TX_RX", name)).read()
        cmd = re.search(r"command=(.*)", txt).group(1).strip()
        rows = []
        for line in txt.splitlines():
            if line.startswith("#") or "|" not in line:          # (rows the reference's CI does not run are commented out in the trace)
                continue
            f = [x.strip() for x in line.replace("||", "|").split("|")]
            rows.append(dict(esn0=float(f[0]), ebn0=float(f[1]), fra=int(f[2]), be=int(f[3]), fe=int(f[4]), ber=float(f[5]), fer=float(f[6]), thr_mbps=float(f[7])))
        hdr = {}
        for key, pat in (("modcod", r"Modulation and coding\s+= (\S+)"), ("max_delay", r"Maximum Channel Delay\s+= (\S+)"), ("implem", r"LDPC implem\s+= (\S+)"),
                         ("n_ite", r"LDPC n iterations\s+= (\d+)"), ("perfect_sync", r"Perfect synchronization = (\S+)"), ("estimator", r"Estimator type\s+= (\S+)")):
            m = re.search(pat, txt)
            hdr[key] = m.group(1) if m else None
        out[name] = dict(command=cmd, header=hdr, rows=rows)
    json.dump(out, open(os.path.join(HERE, "refs_tx_rx.json"), "w"), indent=1)

def src():
    for K in (14232, 9552):          # the rate-8/9 and rate-3/5 payloads (DVBS2.cpp:336-349 names the file per code rate)
        t = open(os.path.join(REF, "conf/src/K_%d.src" % K)).read().split()
        assert t[0] == "1" and t[1] == str(K)
        bits = np.array([int(x) for x in t[2:2 + K]], dtype=np.uint8)
        assert bits.size == K and bits.max() <= 1
        np.save(os.path.join(HERE, "src_K_%d.npy" % K), np.packbits(bits))

def kats():
    from oracle import oracle as O
    from helpers import chain, make_llrs, make_pl_frames
    # LDPC, short 8/9: both schedules
    ch = chain(O, "QPSK-S_8/9")
    _, llr, cw = make_llrs(O, "QPSK-S_8/9", 2, 3.9, seed=101)
    bq, pq, cq, iq = ch.ldpc.decode(llr, 10, 0.875, sched=O.QC, early_stop=False)
    bn, pn, cn, i_n = ch.ldpc.decode(llr, 10, 0.875, sched=O.NATURAL, early_stop=False)
    np.savez_compressed(os.path.join(HERE, "kat_ldpc_short_8_9.npz"), llr=llr, cw=np.packbits(cw.astype(np.uint8), axis=1),
                        bits_qc=np.packbits(bq.astype(np.uint8), axis=1), post_qc=pq, cwd_qc=cq,
                        bits_nat=np.packbits(bn.astype(np.uint8), axis=1), post_nat=pn, cwd_nat=cn,
                        n_ite=10, alpha=0.875)
    # BCH short: codeword + error patterns + expected
    rng = np.random.default_rng(102)
    mc = ch.mc
    info = rng.integers(0, 2, (6, mc.K_bch)).astype(np.int32)
    cwb = ch.bch.encode(info)
    rxb = cwb.copy()
    for f, ne in enumerate([0, 1, 5, 12, 13, 30]):
        rxb[f, rng.choice(mc.N_bch, ne, replace=False)] ^= 1
    out, cwd = ch.bch.decode(rxb)
    np.savez_compressed(os.path.join(HERE, "kat_bch_short.npz"), rx=np.packbits(rxb.astype(np.uint8), axis=1),
                        out=np.packbits(out.astype(np.uint8), axis=1), cwd=cwd, info=np.packbits(info.astype(np.uint8), axis=1),
                        gen=ch.bch.gen().astype(np.uint8))
    # full chain, 16APSK short: PL frame in, info out, intermediate LLRs
    ch2 = chain(O, "16APSK-S_8/9")
    info2, pl, cw2, sigma = make_pl_frames(O, "16APSK-S_8/9", 1, 8.4, seed=103)
    r = ch2.rx(pl[0], sigma=np.float32(sigma), n_ite=10, alpha=0.875, sched=O.QC, early_stop=True)
    np.savez_compressed(os.path.join(HERE, "kat_chain_16apsk_short.npz"), pl=pl[0], sigma=np.float32(sigma),
                        info=np.packbits(info2[0].astype(np.uint8)), llr=r["llr"], out=np.packbits(r["info"].astype(np.uint8)),
                        n_ite=10, alpha=0.875)
    # FIR: taps + input + output (state across two calls)
    taps = O.rrc_taps(0.2, 2, 20)
    x1 = rng.standard_normal(2 * 500).astype(np.float32); x2 = rng.standard_normal(2 * 300).astype(np.float32)
    hist = np.zeros(160, np.float32)
    y1 = O.fir(taps, hist, x1); y2 = O.fir(taps, hist, x2)
    np.savez_compressed(os.path.join(HERE, "kat_fir_rrc81.npz"), taps=taps, x1=x1, x2=x2, y1=y1, y2=y2)

def kats_normal():
    from oracle import oracle as O
    from helpers import chain, make_llrs, make_pl_frames
    ch = chain(O, "QPSK-N_8/9")
    _, l0, c0 = make_llrs(O, "QPSK-N_8/9", 1, 3.0, seed=201)
    _, l1, c1 = make_llrs(O, "QPSK-N_8/9", 1, 4.2, seed=202)
    llr = np.concatenate([l0, l1]); cw = np.concatenate([c0, c1])
    out = dict(llr=llr, cw=np.packbits(cw.astype(np.uint8), axis=1), n_ite=10, alpha=1.0)
    for tag, sched in (("qc", O.QC), ("nat", O.NATURAL)):
        for es in (0, 1):
            b, p_, c, it = ch.ldpc.decode(llr, 10, 1.0, sched=sched, early_stop=bool(es))
            sfx = "%s%s" % (tag, "_es" if es else "")
            out["bits_" + sfx] = np.packbits(b.astype(np.uint8), axis=1); out["cwd_" + sfx] = c; out["ites_" + sfx] = np.asarray(it, np.int32)
            if not es:
                out["post_" + sfx] = p_
    assert out["cwd_qc"].tolist() == [0, 1] and out["ites_qc_es"][1] < 10          # one frame that does not converge, one that does
    np.savez_compressed(os.path.join(HERE, "kat_ldpc_normal_8_9.npz"), **out)
    ch2 = chain(O, "16APSK-N_8/9")
    info2, pl, cw2, sigma = make_pl_frames(O, "16APSK-N_8/9", 1, 8.1, seed=203)
    r = ch2.rx(pl[0], sigma=np.float32(sigma), n_ite=20, alpha=1.0, sched=O.QC, early_stop=False)
    assert np.array_equal(r["info"], info2[0])
    np.savez_compressed(os.path.join(HERE, "kat_chain_16apsk_normal_20ite.npz"), pl=pl[0], sigma=np.float32(sigma),
                        info=np.packbits(info2[0].astype(np.uint8)), llr=r["llr"], out=np.packbits(r["info"].astype(np.uint8)), n_ite=20, alpha=1.0)

def kat_sync():
    """Frame synchronizer (row N4): a noise-free 32APSK-S_3/4 PL stream that starts 321 symbols late."""
    from oracle import oracle as O
    from helpers import chain
    ch = chain(O, "32APSK-S_3/4")
    rng = np.random.default_rng(104)
    F, off = 6, 321
    pl = np.stack([ch.tx(rng.integers(0, 2, ch.mc.K_bch).astype(np.int32))[0] for _ in range(F)])
    n = pl.shape[1] // 2
    stream = np.concatenate([np.zeros(2 * off, np.float32), pl.reshape(-1)])[:F * 2 * n].reshape(F, 2 * n)
    sf = O.SyncFrame(n, alpha=0.9, trigger=30.0, vec_width=8)
    dels, tris, flgs, ys = [], [], [], []
    for f in range(F):
        d, Y = sf.synchronize(stream[f])
        dels.append(d); tris.append(sf.metric); flgs.append(int(sf.packet_flag)); ys.append(Y)
    assert dels[-1] == off and np.array_equal(ys[-1], pl[F - 2])
    np.savez_compressed(os.path.join(HERE, "kat_sync_frame_32apsk.npz"), stream=stream, DEL=np.array(dels, np.int32), TRI=np.array(tris, np.float32),
                        FLG=np.array(flgs, np.int32), Y=np.stack(ys), alpha=0.9, trigger=30.0, vec_width=8)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sync":
        kat_sync()
    elif len(sys.argv) > 1 and sys.argv[1] == "normal":
        kats_normal()
    elif len(sys.argv) > 1 and sys.argv[1] == "refs_full":
        refs_full()
    elif len(sys.argv) > 1 and sys.argv[1] == "src":
        src()
    else:
        pl_seq(); refs(); refs_full(); src(); kats(); kat_sync(); kats_normal()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
