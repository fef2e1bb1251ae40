"""a1 parity: HIP layered-NMS LDPC decoder vs the CPU oracle running the same QC-layer
schedule.  Bar (BASELINE.json north_star): hard decisions bit-exact, soft values within
1e-4 (they are expected to be bit-identical: same fp32 operations in the same order)."""
import numpy as np
import pytest

from helpers import chain, make_llrs, sigma_for

pytestmark = pytest.mark.gpu

LLR_TOL = 1e-4
CASES = [  # modcod, Eb/N0 (dB) near / below the waterfall, frames
    ("QPSK-S_8/9", 4.6, 6), ("QPSK-S_8/9", 3.0, 4),
    ("QPSK-S_3/5", 2.4, 4), ("QPSK-S_3/5", 0.8, 3),
    ("32APSK-S_3/4", 3.4, 4),
    ("QPSK-N_8/9", 4.2, 3), ("QPSK-N_8/9", 3.0, 2),
]


@pytest.fixture(scope="module")
def Rx():
    from dvbs2_amd.receiver import Dvbs2Hip
    return Dvbs2Hip


@pytest.mark.parametrize("modcod,ebn0,F", CASES)
@pytest.mark.parametrize("early", [False, True])
def test_ldpc_matches_oracle(O, Rx, modcod, ebn0, F, early):
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=11)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=0.875, early_stop=early)
    V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=10, alpha=0.875, sched=O.QC, early_stop=early)
    assert np.array_equal(V, Vo), "hard decisions differ: %d bits" % int((V != Vo).sum())
    assert np.array_equal(CWD, cwdo)
    assert np.array_equal(ites, iteso)
    assert np.max(np.abs(post - posto)) <= LLR_TOL
    assert np.array_equal(post, posto), "posteriors are expected to be bit-identical"
    rx.close()


@pytest.mark.parametrize("alpha", [1.0, 0.75])
def test_ldpc_alpha_and_iterations(O, Rx, alpha):
    modcod = "QPSK-S_8/9"
    ch = chain(O, modcod)
    _, llr, _ = make_llrs(O, modcod, 3, 3.8, seed=5)
    rx = Rx(modcod, max_frames=3, n_ite=1, alpha=alpha, early_stop=False)
    for n_ite in (1, 2, 7, 20):
        rx.set_ldpc_params(n_ite, alpha, False)
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=n_ite, alpha=alpha, sched=O.QC, early_stop=False)
        assert np.array_equal(V, Vo) and np.array_equal(post, posto) and np.array_equal(CWD, cwdo)
        assert (ites == n_ite).all()
    rx.close()


def test_ldpc_hybrid_storage_policies_agree(O, Rx, monkeypatch):
    """Where the posteriors live (LDS vs per-workgroup global workspace) and where the packed
    c->v state lives must not change a single bit."""
    modcod = "QPSK-S_8/9"
    ch = chain(O, modcod)
    _, llr, _ = make_llrs(O, modcod, 4, 4.0, seed=7)
    Vo, posto, _, _ = ch.ldpc.decode(llr, n_ite=6, alpha=1.0, sched=O.QC, early_stop=False)
    for c2v in ("lds", "global"):
        for groups in (0, 7, 30, -1):
            monkeypatch.setenv("DVBS2HIP_LDPC_C2V", c2v)
            rx = Rx(modcod, max_frames=4, n_ite=6, alpha=1.0, early_stop=False, lds_groups=groups)
            V, _, post, _ = rx.decode_siho(llr, with_post=True)
            assert np.array_equal(V, Vo) and np.array_equal(post, posto), (c2v, groups)
            rx.close()


def test_ldpc_batch_larger_than_grid_and_roundtrip(O, Rx):
    """More frames than resident workgroups (persistent grid loops), noiseless and noisy:
    encode -> decode round trip is the identity; every frame of the batch is decoded."""
    modcod = "QPSK-S_8/9"
    ch = chain(O, modcod)
    F = 1500
    rng = np.random.default_rng(3)
    info = rng.integers(0, 2, (8, ch.mc.K_ldpc)).astype(np.int32)
    cw = ch.ldpc.encode(info)
    idx = rng.integers(0, 8, F)
    sigma = 0.28
    y = (1.0 - 2.0 * cw[idx]) + sigma * rng.standard_normal((F, ch.mc.N_ldpc))
    llr = (2 * y / sigma ** 2).astype(np.float32)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=True)
    V, CWD = rx.decode_siho(llr)
    assert np.array_equal(V, info[idx])
    assert (CWD == 1).all()
    rx.close()


def test_ldpc_normal_frame_full_batch_properties(O, Rx):
    """BASELINE config 2 shape (N=64800, 8/9): size-independent properties on a batch the
    oracle could not finish: round trip at high SNR, all-zero codeword symmetry (decoding
    llr with the codeword's signs flipped gives the all-zero word), determinism."""
    modcod = "QPSK-N_8/9"
    ch = chain(O, modcod)
    mc = ch.mc
    F = 512
    rng = np.random.default_rng(9)
    info = rng.integers(0, 2, (4, mc.K_ldpc)).astype(np.int32)
    cw = ch.ldpc.encode(info)
    idx = rng.integers(0, 4, F)
    sigma = 0.30
    noise = sigma * rng.standard_normal((F, mc.N_ldpc))
    llr = (2 * ((1.0 - 2.0 * cw[idx]) + noise) / sigma ** 2).astype(np.float32)
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
    V, CWD = rx.decode_siho(llr)
    assert np.array_equal(V, info[idx]) and (CWD == 1).all()
    V2, _ = rx.decode_siho(llr)
    assert np.array_equal(V, V2)
    # symmetry: flip LLR signs where the codeword bit is 1 -> the decoder must output zeros
    llr0 = (llr * (1.0 - 2.0 * cw[idx])).astype(np.float32)
    V0, C0 = rx.decode_siho(llr0)
    assert not V0.any() and (C0 == 1).all()
    # the first 3 frames agree with the oracle bit for bit
    Vo, _, _, _ = ch.ldpc.decode(llr[:3], n_ite=10, alpha=1.0, sched=O.QC, early_stop=False)
    assert np.array_equal(V[:3], Vo)
    rx.close()


@pytest.mark.parametrize("modcod", ["QPSK-S_8/9", "QPSK-N_8/9", "32APSK-S_3/4"])
def test_ldpc_ties_zeros_and_extremes_match_oracle(O, Rx, modcod):
    """Edge cases of the check-node rule: every |LLR| equal (min1 == min2 ties everywhere: the
    packed state keeps ONE minimum position, the oracle compares values), exact zeros (sign of
    zero, -0.0 messages), huge magnitudes, an all-zero frame."""
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(33)
    info = rng.integers(0, 2, (1, mc.K_ldpc)).astype(np.int32)
    cw = ch.ldpc.encode(info)[0]
    s = (1.0 - 2.0 * cw).astype(np.float32)
    frames = []
    x = s.copy(); x[rng.choice(mc.N_ldpc, mc.N_ldpc // 400, replace=False)] *= -1; frames.append(x)           # hard +-1: ties
    x = s.copy() * 2.5; x[rng.choice(mc.N_ldpc, mc.N_ldpc // 100, replace=False)] = 0.0; frames.append(x)     # erasures (exact 0)
    x = s.copy() * 3e4; x[rng.choice(mc.N_ldpc, mc.N_ldpc // 50, replace=False)] *= -1e-3; frames.append(x)   # huge dynamic range
    frames.append(np.zeros(mc.N_ldpc, np.float32))                                                            # nothing received
    x = s.copy(); x[::2] = -0.0; frames.append(x)                                                             # negative zeros
    llr = np.stack(frames)
    F = llr.shape[0]
    for early in (False, True):
        rx = Rx(modcod, max_frames=F, n_ite=8, alpha=0.75, early_stop=early)
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=8, alpha=0.75, sched=O.QC, early_stop=early)
        assert np.array_equal(V, Vo) and np.array_equal(CWD, cwdo) and np.array_equal(ites, iteso)
        assert np.array_equal(post.view(np.uint32), posto.view(np.uint32)), "bit patterns (incl. the sign of zero) must match"
        rx.close()
    assert np.array_equal(V[1], info[0])      # 1 % erasures are filled in


@pytest.mark.parametrize("implem", ["NMS", "SPA", "SPA_TANH", "SPA_EXACT"])
@pytest.mark.parametrize("modcod", ["QPSK-S_8/9", "QPSK-N_8/9"])
def test_ldpc_non_finite_llrs_end_and_stay_in_their_frame(O, Rx, modcod, implem):
    """+-inf and NaN on the LLR socket (a demapper fed sigma = 0, a saturated front end): the call has to come back (every loop of the kernels is bounded by n_ite), frames WITHOUT
    such values have to come out bit for bit as they do when decoded alone (frames share a workgroup's queue, nothing else), and a frame whose infinities all carry the right sign is
    decoded: certain bits can only help."""
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(77)
    info = rng.integers(0, 2, (4, mc.K_ldpc)).astype(np.int32)
    cw = ch.ldpc.encode(info)
    sigma = sigma_for(mc, 4.6)
    llr = ((1.0 - 2.0 * cw) + sigma * rng.standard_normal(cw.shape)).astype(np.float32) * np.float32(2.0 / sigma ** 2)
    clean = llr.copy()
    pos = rng.choice(mc.N_ldpc, mc.N_ldpc // 20, replace=False)
    llr[1, pos] = np.where(cw[1, pos] == 0, np.inf, -np.inf).astype(np.float32)          # certain and right
    llr[2, pos[:50]] = np.nan
    llr[2, pos[50:100]] = np.where(cw[2, pos[50:100]] == 0, -np.inf, np.inf)             # certain and wrong
    for early in (False, True):
        rx = Rx(modcod, max_frames=4, n_ite=10, alpha=0.875, early_stop=early, implem=implem)
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        V0, CWD0, post0, ites0 = rx.decode_siho(clean[[0, 3]], with_post=True)
        rx.close()
        assert np.array_equal(V[[0, 3]], V0) and np.array_equal(CWD[[0, 3]], CWD0) and np.array_equal(ites[[0, 3]], ites0)
        assert np.array_equal(post[[0, 3]].view(np.uint32), post0.view(np.uint32))
        assert np.array_equal(V[1], info[1]) and CWD[1] == 1
        assert np.array_equal(V[0], info[0]) and np.array_equal(V[3], info[3])
        assert ites.min() >= 1 and ites.max() <= 10


SPA_RULES = [("SPA", "SPA_CLIP"), ("SPA_EXACT", "SPA")]      # (--dec-implem, the oracle's rule): SPA = the exact check node with AFF3CT's message cap, SPA_EXACT = without


@pytest.mark.parametrize("implem,orule", SPA_RULES)
@pytest.mark.parametrize("modcod,ebn0", [("QPSK-S_8/9", 3.9), ("QPSK-S_3/5", 1.8), ("32APSK-S_3/4", 3.0), ("QPSK-N_8/9", 3.9)])
def test_ldpc_spa_matches_oracle(O, Rx, modcod, ebn0, implem, orule):
    """--dec-implem SPA (the reference's default) / SPA_EXACT: exact boxplus check node, with / without the cap at 2 atanh(1 - FLT_EPSILON).  The GPU evaluates the
    exp/log terms on the hardware exp2/log2 units, the oracle with libm, so the parity bar is the
    soft one: |posterior difference| <= 1e-4 * max(1, |posterior|) after 1 and 2 iterations (before
    rounding differences can be amplified), and identical hard decisions / iteration counts for
    frames that converge."""
    ch = chain(O, modcod)
    F = 4
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=17)
    for n_ite in (1, 2):
        rx = Rx(modcod, max_frames=F, n_ite=n_ite, early_stop=False, implem=implem)
        V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=n_ite, implem=getattr(O, orule), sched=O.QC, early_stop=False)
        assert np.all(np.abs(post - posto) <= 1e-4 * np.maximum(1.0, np.abs(posto))), float(np.abs(post - posto).max())
        assert (V != Vo).mean() < 1e-4
        rx.close()
    rx = Rx(modcod, max_frames=F, n_ite=50, early_stop=True, implem=implem)
    V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=50, implem=getattr(O, orule), sched=O.QC, early_stop=True)
    conv = (CWD == 1) & (cwdo == 1)
    assert conv.sum() >= F - 1
    assert np.array_equal(V[conv], Vo[conv]) and np.all(np.abs(ites[conv] - iteso[conv]) <= 1)
    assert np.array_equal(V[conv], cw[conv][:, :ch.mc.K_ldpc])
    rx.close()


TANH_CASES = [("QPSK-S_8/9", 3.9, 4), ("QPSK-S_8/9", 3.2, 3), ("QPSK-S_3/5", 1.5, 4), ("8PSK-S_3/5", 3.0, 3), ("16APSK-S_8/9", 7.4, 3), ("32APSK-S_3/4", 3.2, 3), ("QPSK-N_8/9", 3.9, 2)]


@pytest.mark.parametrize("modcod,ebn0,F", TANH_CASES)
def test_ldpc_spa_tanh_is_the_oracles_rule_bit_for_bit(O, Rx, modcod, ebn0, F):
    """--dec-implem SPA_TANH: the check node as AFF3CT's Update_rule_SPA evaluates it (tanh product in fp32, quotient clamped to 1 - FLT_EPSILON, 2 atanh) -- messages
    cap at 16.64 and, near the cap, move in steps of 0.1 .. 0.7, so "within 1e-4" is not a bar a twin of the oracle can be held to: both sides are written with
    correctly rounded operations only (oracle/dvbs2_oracle.c chk_update_spa_tanh, k_ldpc_wg8.hip w8_det_*), the kernel multiplies a check's tanh values in the
    oracle's edge order, and the bar is the min-sum family's: posteriors BIT-IDENTICAL after 1, 2 and 50 iterations (every frame of the 50-iteration runs is deep
    in the saturated regime), same iteration counts with the stopping rule."""
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=31)
    for n_ite, early in ((1, False), (2, False), (50, False), (50, True)):
        rx = Rx(modcod, max_frames=F, n_ite=n_ite, early_stop=early, implem="SPA_TANH")
        assert rx.ldpc_kernel_name().endswith(",2>")
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=n_ite, implem=O.SPA_TANH, sched=O.QC, early_stop=early)
        assert not np.isnan(post).any()
        assert np.array_equal(post.view(np.uint32), posto.view(np.uint32)), "%d posteriors differ after %d iterations, max %g" % (int((post.view(np.uint32) != posto.view(np.uint32)).sum()), n_ite, float(np.abs(post - posto).max()))
        assert np.array_equal(V, Vo) and np.array_equal(CWD, cwdo) and np.array_equal(ites, iteso)
        if n_ite == 50 and not early:
            assert float(np.abs(posto).max()) > 20.0          # beyond one capped message: several saturated messages per bit
        rx.close()


def test_ldpc_spa_tanh_saturates_where_the_exact_rule_does_not(O, Rx):
    """What the two sum-product rules differ by: scaled-up LLRs drive the exact rule's posteriors to thousands, the tanh-product rule's messages stop at
    2 atanh(1 - 2^-23) = 16.64 each -- a bit of column weight w cannot exceed |channel| + 16.64 w."""
    modcod = "QPSK-S_8/9"
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, 2, 9.0, seed=23)
    llr = (llr * 4.0).astype(np.float32)
    rx = Rx(modcod, max_frames=2, n_ite=20, early_stop=False, implem="SPA_TANH")
    V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=20, implem=O.SPA_TANH, sched=O.QC, early_stop=False)
    assert np.array_equal(post.view(np.uint32), posto.view(np.uint32)) and CWD.all()
    assert np.all(np.abs(post) <= np.abs(llr) + 16.636 * 4 + 1e-3)           # (the columns of the short 8/9 code have at most 4 edges: SURVEY 7 H1)
    rx.close()


@pytest.mark.parametrize("modcod,ebn0,n_ite,scale", [("QPSK-S_8/9", 5.0, 10, 1.0), ("QPSK-S_8/9", 9.0, 20, 4.0), ("QPSK-S_3/5", 6.0, 10, 1.0), ("32APSK-S_3/4", 8.0, 10, 1.0),
                                                     ("QPSK-N_8/9", 6.0, 10, 1.0), ("QPSK-N_8/9", 9.0, 10, 3.0)])
def test_ldpc_spa_far_beyond_the_saturation_of_tanh(O, Rx, modcod, ebn0, n_ite, scale):
    """The sum-product kernel evaluates the check node in the complement-product domain with a per-check scale (k_ldpc_wg8.hip): fixed iterations far
    above the waterfall drive the posteriors to hundreds and thousands (11 800 in the second case) -- where tanh(a / 2) is 1 in fp32, e^-a leaves
    the fp32 range, and a check's weakest edge can lie 60 and more below its second weakest -- and the GPU still has to follow the oracle's boxplus
    recursions: posteriors within 1e-4 max(1, abs(L)) (measured: 3e-6), no NaN, identical hard decisions and CWD."""
    ch = chain(O, modcod)
    F = 2
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=23)
    llr = (llr * scale).astype(np.float32)
    rx = Rx(modcod, max_frames=F, n_ite=n_ite, early_stop=False, implem="SPA_EXACT")
    V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=n_ite, implem=O.SPA, sched=O.QC, early_stop=False)
    assert not np.isnan(post).any() and np.isfinite(post).all()
    assert float(np.abs(posto).max()) > 400.0                                    # the regime the test is about
    assert np.all(np.abs(post - posto) <= 1e-4 * np.maximum(1.0, np.abs(posto))), float((np.abs(post - posto) / np.maximum(1.0, np.abs(posto))).max())
    assert np.array_equal(V, Vo) and np.array_equal(CWD, cwdo) and CWD.all()
    rx.close()


@pytest.mark.parametrize("modcod,ebn0,n_ite,scale", [("QPSK-S_8/9", 9.0, 20, 4.0), ("QPSK-S_3/5", 6.0, 10, 1.0), ("QPSK-N_8/9", 9.0, 10, 3.0)])
def test_ldpc_spa_default_clips_where_the_reference_saturates(O, Rx, modcod, ebn0, n_ite, scale):
    """--dec-implem SPA in the same regime: every c->v message stops at 16.6355 = 2 atanh(1 - FLT_EPSILON), the cap of AFF3CT's fp32 tanh product (the oracle's ORC_SPA_CLIP:
    exact boxplus, then the clip), so a posterior cannot leave |channel| + 16.6355 x column weight; same bar as the unclipped rule, 1e-4 max(1, |L|)."""
    ch = chain(O, modcod)
    F = 2
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=23)
    llr = (llr * scale).astype(np.float32)
    rx = Rx(modcod, max_frames=F, n_ite=n_ite, early_stop=False, implem="SPA")
    V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=n_ite, implem=O.SPA_CLIP, sched=O.QC, early_stop=False)
    assert np.isfinite(post).all()
    assert np.all(np.abs(post - posto) <= 1e-4 * np.maximum(1.0, np.abs(posto))), float((np.abs(post - posto) / np.maximum(1.0, np.abs(posto))).max())
    assert np.all(np.abs(post) <= np.abs(llr) + 16.6356 * 13 + 1e-2) and float(np.abs(post - llr).max()) > 3 * 16.0      # (column weights: at most 13 on these codes)
    assert np.array_equal(V, Vo) and np.array_equal(CWD, cwdo) and CWD.all()
    rx.close()


NAT_CASES = [("QPSK-S_8/9", 4.4, 5), ("QPSK-S_8/9", 3.2, 70), ("QPSK-S_3/5", 1.2, 3), ("32APSK-S_3/4", 3.4, 4), ("QPSK-N_8/9", 3.6, 2)]


@pytest.mark.parametrize("modcod,ebn0,F", NAT_CASES)
@pytest.mark.parametrize("early", [False, True])
@pytest.mark.parametrize("parts", [1, 4, 8, 88, 44, 82, 81, 41, 48])
def test_ldpc_natural_order_matches_oracle(O, Rx, monkeypatch, modcod, ebn0, F, early, parts):
    """dvbs2hip_set_ldpc_schedule(NATURAL): the reference's sweep order (checks in row order) against the oracle's ORC_SCHED_NATURAL -- hard decisions, iteration
    counts and posteriors bit for bit -- in its forms: one lane per frame (64 frames per wave), a check's edges split over 4 / 8 adjacent lanes (16 / 8 frames per
    wave: the merged minima, the sign word and the forwarded parity posterior come from exchanges inside the wave), and 8 / 4 CONSECUTIVE checks of a frame in adjacent
    lanes (two digits = checks side by side, waves per workgroup: the parity chain as a scan over the lanes, the waves of a workgroup sharing the image's rows of 8 to 64 frames;
    82 is the default up to 16 frames per CU, 44 beyond); more frames than one wave or workgroup holds, a ragged last group."""
    from dvbs2_amd import lib_binding as B
    monkeypatch.setenv("DVBS2HIP_NAT_PARTS", str(parts))
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=21)
    rx = Rx(modcod, max_frames=F, n_ite=8, alpha=0.875, early_stop=early)
    rx.set_ldpc_schedule(B.SCHED_NATURAL)
    assert rx.ldpc_kernel_name().startswith("ldpc_nat_kernel")
    V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=8, alpha=0.875, sched=O.NATURAL, early_stop=early)
    assert np.array_equal(V, Vo), "hard decisions differ: %d bits" % int((V != Vo).sum())
    assert np.array_equal(CWD, cwdo) and np.array_equal(ites, iteso)
    assert np.array_equal(post, posto), "posteriors are expected to be bit-identical (max diff %g)" % float(np.max(np.abs(post - posto)))
    # and back: the same handle on the QC schedule
    rx.set_ldpc_schedule(B.SCHED_QC)
    V2, _, _, _ = rx.decode_siho(llr, with_post=True)
    Vq, _, _, _ = ch.ldpc.decode(llr, n_ite=8, alpha=0.875, sched=O.QC, early_stop=early)
    assert np.array_equal(V2, Vq)
    rx.close()


@pytest.mark.parametrize("modcod,ebn0,F", [("QPSK-S_8/9", 3.9, 5), ("QPSK-S_8/9", 3.2, 70), ("QPSK-S_3/5", 1.5, 3), ("8PSK-S_3/5", 3.0, 3), ("32APSK-S_3/4", 3.2, 3), ("QPSK-N_8/9", 3.9, 2)])
def test_ldpc_sum_product_in_the_references_sweep_order(O, Rx, modcod, ebn0, F):
    """The reference's decoder as recalled -- Decoder_LDPC_BP_horizontal_layered<.., Update_rule_SPA>: the tanh-product check node, the checks in the row order of H -- is the
    oracle's orc_ldpc_decode(ORC_SPA_TANH, ORC_SCHED_NATURAL); on the device: --dec-implem SPA_TANH with dvbs2hip_set_ldpc_schedule(NATURAL) (ldpc_nat_spa_kernel<.., 2>: one
    lane per frame, every operation correctly rounded and in the oracle's order), BIT FOR BIT after 1, 2 and 50 iterations and with the stopping rule.  `SPA` (exact check
    node + AFF3CT's cap) and SPA_EXACT in the same sweep order: within 1e-4 max(1, |L|) of the oracle after 1 and 2 iterations, converged frames = the sent word."""
    from dvbs2_amd import lib_binding as B
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=37)
    for n_ite, early in ((1, False), (2, False), (50, False), (50, True)):
        rx = Rx(modcod, max_frames=F, n_ite=n_ite, early_stop=early, implem="SPA_TANH")
        rx.set_ldpc_schedule(B.SCHED_NATURAL)
        assert rx.ldpc_kernel_name() == "ldpc_nat_spa_kernel<%d,2>" % (27 if "8/9" in modcod else 11 if "3/5" in modcod else 13)
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=n_ite, implem=O.SPA_TANH, sched=O.NATURAL, early_stop=early)
        assert np.array_equal(post.view(np.uint32), posto.view(np.uint32)), "%d posteriors differ after %d iterations" % (int((post.view(np.uint32) != posto.view(np.uint32)).sum()), n_ite)
        assert np.array_equal(V, Vo) and np.array_equal(CWD, cwdo) and np.array_equal(ites, iteso)
        rx.close()
    for implem, orule in SPA_RULES:
        for n_ite in (1, 2):
            rx = Rx(modcod, max_frames=F, n_ite=n_ite, early_stop=False, implem=implem)
            rx.set_ldpc_schedule(B.SCHED_NATURAL)
            V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
            Vo, posto, cwdo, _ = ch.ldpc.decode(llr, n_ite=n_ite, implem=getattr(O, orule), sched=O.NATURAL, early_stop=False)
            assert np.all(np.abs(post - posto) <= 1e-4 * np.maximum(1.0, np.abs(posto))), float(np.abs(post - posto).max())
            assert (V != Vo).mean() < 1e-4
            rx.close()
        rx = Rx(modcod, max_frames=F, n_ite=50, early_stop=True, implem=implem)
        rx.set_ldpc_schedule(B.SCHED_NATURAL)
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=50, implem=getattr(O, orule), sched=O.NATURAL, early_stop=True)
        conv = (CWD == 1) & (cwdo == 1)
        assert conv.sum() >= cwdo.sum() - 1 and np.array_equal(V[conv], cw[conv][:, :ch.mc.K_ldpc]) and np.all(np.abs(ites[conv] - iteso[conv]) <= 1)      # (3.2 dB: no frame converges on either side)
        rx.close()


def test_natural_order_in_the_fused_chain(O, Rx):
    from dvbs2_amd import lib_binding as B
    from helpers import make_pl_frames
    modcod = "16APSK-S_8/9"
    info, pl, _, sigma = make_pl_frames(O, modcod, 3, 8.6, seed=22)
    rx = Rx(modcod, max_frames=3, n_ite=10, alpha=1.0, early_stop=True)
    rx.set_ldpc_schedule(B.SCHED_NATURAL)
    out, c0, c1 = rx.rx_bb(pl, sigma=np.float32(sigma))
    ch = chain(O, modcod)
    for f in range(3):
        r = ch.rx(pl[f], sigma=np.float32(sigma), n_ite=10, alpha=1.0, sched=O.NATURAL, early_stop=True)
        assert np.array_equal(out[f], r["info"])
    assert np.array_equal(out, info)
    rx.close()


DEFAULT_NMS_MODE = 5       # image mode the planner picks for the min-sum decoder on normal frames (6 = one frame per CU, k_ldpc_cu1.hip)
DEFAULT_SPA_MODE = 6       # ... for the sum-product decoder (round 5: one frame per CU, two lanes per check, the messages inside the Infinity Cache)


def _big_batch(O, modcod, F, ebn0s, seed, n_cw=8):
    """F channel-LLR frames (BPSK-equivalent) of n_cw oracle-encoded codewords, frame f at ebn0s[f % len]"""
    ch = chain(O, modcod)
    mc = ch.mc
    rng = np.random.default_rng(seed)
    info = rng.integers(0, 2, (n_cw, mc.K_bch)).astype(np.int32)
    bch = ch.bch.encode(info)
    cw = ch.ldpc.encode(bch)
    idx = rng.integers(0, n_cw, F)
    rate = mc.K_bch / mc.N_ldpc
    eb = np.asarray(ebn0s, np.float64)[np.arange(F) % len(ebn0s)]
    sigma = np.sqrt(1.0 / (2.0 * rate * 10.0 ** (eb / 10.0))).astype(np.float32)[:, None]
    llr = np.empty((F, mc.N_ldpc), np.float32)
    for s0 in range(0, F, 256):
        e = min(F, s0 + 256)
        y = (1.0 - 2.0 * cw[idx[s0:e]]).astype(np.float32) + sigma[s0:e] * rng.standard_normal((e - s0, mc.N_ldpc), dtype=np.float32)
        llr[s0:e] = 2.0 * y / sigma[s0:e] ** 2
    return ch, bch[idx], llr


@pytest.mark.parametrize("n_ite,F", [(10, 2304), (20, 1536)])
@pytest.mark.parametrize("ebn0s", [(3.0,), (3.0, 4.2)])
def test_ldpc_headline_config_at_size_matches_oracle(O, Rx, n_ite, F, ebn0s):
    """BASELINE configs[1] (N = 64800 8/9, 10 ite) and configs[3]'s 20 iterations AT SIZE: more frames than the persistent grid
    holds (2 x 256 workgroups), so every workgroup decodes a 2nd .. 5th frame through the work queue on an image and a c->v state
    that the previous frame left behind, at 3.0 dB where frames do NOT converge (stale state would show) and in a 3.0 / 4.2 dB mix
    where frames of different length interleave under the early stop.  The LAST 8 frames and 8 random ones are compared with the
    oracle: posterior bit patterns, hard decisions, CWD, iteration counts; fixed iterations and early stop."""
    modcod = "QPSK-N_8/9"
    ch, sent, llr = _big_batch(O, modcod, F, ebn0s, seed=1000 + n_ite + len(ebn0s))
    rng = np.random.default_rng(5)
    pick = np.unique(np.concatenate([np.arange(F - 8, F), rng.choice(F - 8, 8, replace=False)]))
    for early in (False, True):
        rx = Rx(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=early)
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr[pick], n_ite=n_ite, alpha=1.0, sched=O.QC, early_stop=early)
        assert np.array_equal(post[pick].view(np.uint32), posto.view(np.uint32)), (early, "posterior bit patterns")
        assert np.array_equal(V[pick], Vo) and np.array_equal(CWD[pick], cwdo) and np.array_equal(ites[pick], iteso)
        # size-independent properties over the whole batch: a detected codeword is the sent one; the easy half converges
        okf = CWD == 1
        assert np.array_equal(V[okf], sent[okf])
        if len(ebn0s) == 2:
            assert okf[1::2].mean() > 0.99 and (ites[1::2] < n_ite).mean() > (0.9 if early else -1)
        if not early:
            assert (ites == n_ite).all()
            V2, CWD2 = rx.decode_siho(llr)                  # same launch again: nothing of the first one survives in the workspaces
            assert np.array_equal(V, V2) and np.array_equal(CWD, CWD2)
        rx.close()
    assert (CWD[0::2] if len(ebn0s) == 2 else CWD).mean() < 0.5      # 3.0 dB: most frames fail, which is the point


@pytest.mark.parametrize("modcod,ebn0", [("QPSK-S_8/9", 4.2), ("QPSK-S_3/5", 2.0), ("32APSK-S_3/4", 3.4)])
def test_two_lanes_per_check_kernel_is_bit_exact(O, Rx, monkeypatch, modcod, ebn0):
    """k_ldpc_lat.hip (opt-in, DVBS2HIP_LDPC_LAT=1: written for the one-frame call's latency, measured slower than the lone workgroup it was meant to beat): one frame per CU, a
    check's slots over two adjacent lanes, image and packed state in LDS -- posteriors, hard decisions, CWD and iteration counts bit for bit the oracle's QC schedule, the
    27-, 11- and padded 13-slot layers, with and without the stopping rule."""
    monkeypatch.setenv("DVBS2HIP_LDPC_LAT", "1")
    ch = chain(O, modcod)
    F = 5
    _, llr, cw = make_llrs(O, modcod, F, ebn0, seed=71)
    for early, alpha in ((False, 1.0), (True, 0.875)):
        rx = Rx(modcod, max_frames=F, n_ite=10, alpha=alpha, early_stop=early)
        assert rx.ldpc_kernel_name().startswith("ldpc_lat_kernel<")
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=10, alpha=alpha, sched=O.QC, early_stop=early)
        assert np.array_equal(post.view(np.uint32), posto.view(np.uint32)) and np.array_equal(V, Vo) and np.array_equal(CWD, cwdo) and np.array_equal(ites, iteso)
        rx.close()


def test_small_batch_handles_take_the_one_frame_per_cu_image(O, Rx):
    """(round 6) A handle created for at most one frame per CU (max_frames <= the CU count) decodes normal frames with the one-frame-per-CU image
    (k_ldpc_cu1.hip: two lanes per check -- a call is one frame's ten iterations on one CU, 0.42 ms instead of 0.54); a larger handle keeps the two-frames-per-CU kernel.
    Same decoder either way: hard decisions, CWD, iteration counts and posteriors bit for bit, against each other and against the oracle, with and without the stopping rule."""
    modcod, F = "QPSK-N_8/9", 6
    ch = chain(O, modcod)
    _, llr, cw = make_llrs(O, modcod, F, 3.9, seed=61)
    for early in (False, True):
        small = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=early)
        big = Rx(modcod, max_frames=1024, n_ite=10, alpha=1.0, early_stop=early)
        assert small.ldpc_kernel_name() == "ldpc_cu1_kernel<27>" and big.ldpc_kernel_name() == "ldpc_wg8_kernel<27,%d>" % DEFAULT_NMS_MODE
        a, b = small.decode_siho(llr, with_post=True), big.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr, n_ite=10, alpha=1.0, sched=O.QC, early_stop=early)
        for x, y, z in zip(a, b, (Vo, cwdo, posto, iteso)):
            v = (lambda t: t.view(np.uint32) if t.dtype == np.float32 else t)
            assert np.array_equal(v(x), v(y)) and np.array_equal(v(x), v(z))
        small.close(); big.close()


def test_ldpc_baseline_batch_of_exactly_4096_frames_matches_oracle(O, Rx):
    """BASELINE configs[1] AT ITS STATED SIZE: 4096 QPSK-N 8/9 frames in ONE call, 10 fixed iterations -- the batch bench.py times -- at 3.0 dB (nothing
    converges) and 4.2 dB interleaved: 16 random frames + the last 8 against the oracle by posterior BIT PATTERN, hard decisions and CWD; then the same batch in the
    reference's own sweep order (SCHED_NATURAL, AFF3CT's BP_HORIZONTAL_LAYERED over the rows of H as built, DVBS2.cpp:428), 8 frames against the oracle's natural schedule.
    Size-independent over the whole batch: a detected codeword is the sent word, the 4.2 dB half decodes, the 3.0 dB half does not."""
    modcod, F = "QPSK-N_8/9", 4096
    ch, sent, llr = _big_batch(O, modcod, F, (3.0, 4.2), seed=4096, n_cw=8)
    rng = np.random.default_rng(40)
    pick = np.unique(np.concatenate([np.arange(F - 8, F), rng.choice(F - 8, 16, replace=False)]))
    rx = Rx(modcod, max_frames=F, n_ite=10, alpha=1.0, early_stop=False)
    assert rx.ldpc_kernel_name() == "ldpc_wg8_kernel<27,%d>" % DEFAULT_NMS_MODE
    V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, iteso = ch.ldpc.decode(llr[pick], n_ite=10, alpha=1.0, sched=O.QC, early_stop=False)
    assert np.array_equal(post[pick].view(np.uint32), posto.view(np.uint32)), "posterior bit patterns (QC-layer order)"
    assert np.array_equal(V[pick], Vo) and np.array_equal(CWD[pick], cwdo) and (ites == 10).all()
    okf = CWD == 1
    assert np.array_equal(V[okf], sent[okf]) and okf[1::2].mean() > 0.99 and okf[0::2].mean() < 0.05
    Vb, CWDb = rx.decode_siho(llr)                      # the bits socket alone (the bench line's step): same decisions
    assert np.array_equal(Vb, V) and np.array_equal(CWDb, CWD)
    from dvbs2_amd import lib_binding as B
    rx.set_ldpc_schedule(B.SCHED_NATURAL)
    pick_n = pick[-8:]
    Vn, CWDn, postn, _ = rx.decode_siho(llr, with_post=True)
    Vno, postno, cwdno, _ = ch.ldpc.decode(llr[pick_n], n_ite=10, alpha=1.0, sched=O.NATURAL, early_stop=False)
    assert np.array_equal(postn[pick_n].view(np.uint32), postno.view(np.uint32)), "posterior bit patterns (natural row order)"
    assert np.array_equal(Vn[pick_n], Vno) and np.array_equal(CWDn[pick_n], cwdno)
    okn = CWDn == 1
    assert np.array_equal(Vn[okn], sent[okn]) and okn[1::2].mean() > 0.99
    rx.close()


@pytest.mark.parametrize("implem", ["NMS", "SPA", "SPA_EXACT"])
@pytest.mark.parametrize("mode,kernel_mode", [("", None), ("cu1", 6), ("park", 5), ("park4", 4), ("static", 3), ("global", 1)])
def test_ldpc_normal_frame_image_modes_agree_with_the_oracle(O, Rx, monkeypatch, mode, kernel_mode, implem):
    """Where the posteriors of a normal frame live -- static hybrid with 39 / 32 bit-group rows parked in the idle waves' registers (modes 5 / 4, the
    defaults of the min-sum / sum-product kernel), static hybrid alone (3), the workgroup's global slot (1) -- must not change a bit: 1100 frames
    (every persistent workgroup decodes a 2nd and 3rd frame; the parked rows' swap schedule runs through iterations, syndrome sweeps that stop
    early and frames that never converge) at 3.0 / 4.2 dB, fixed iterations and the stopping rule, against the oracle on the last 6 + 6 random frames."""
    modcod, F = "QPSK-N_8/9", 1100
    if mode:
        monkeypatch.setenv("DVBS2HIP_LDPC_FAST_MODE", mode)
    ch, sent, llr = _big_batch(O, modcod, F, (3.0, 4.2), seed=4242, n_cw=4)
    rng = np.random.default_rng(8)
    pick = np.unique(np.concatenate([np.arange(F - 6, F), rng.choice(F - 6, 6, replace=False)]))
    spa = implem != "NMS"
    orule = {"NMS": O.NMS, "SPA": O.SPA_CLIP, "SPA_EXACT": O.SPA}[implem]
    for early in (False, True):
        n_ite = 3 if spa and not early else 10
        rx = Rx(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=early, implem=implem)
        want = kernel_mode if kernel_mode is not None else (DEFAULT_SPA_MODE if spa else DEFAULT_NMS_MODE)
        if spa and want == 5:
            want = 4                    # (the sum-product kernel has no registers for 39 parked rows)
        name = {"NMS": "ldpc_cu1_kernel<27>", "SPA": "ldpc_cu1_kernel<27,3>", "SPA_EXACT": "ldpc_cu1_kernel<27,1>"}[implem] if want == 6 else "ldpc_wg8_kernel<27,%d%s>" % (want, {"NMS": "", "SPA": ",3", "SPA_EXACT": ",1"}[implem])
        assert rx.ldpc_kernel_name() == name, rx.ldpc_kernel_name()
        V, CWD, post, ites = rx.decode_siho(llr, with_post=True)
        Vo, posto, cwdo, iteso = ch.ldpc.decode(llr[pick], n_ite=n_ite, alpha=1.0, implem=orule, sched=O.QC, early_stop=early)
        if spa:
            assert np.all(np.abs(post[pick] - posto) <= 1e-4 * np.maximum(1.0, np.abs(posto))), float(np.abs(post[pick] - posto).max())
            assert (V[pick] != Vo).mean() < 1e-4
        else:
            assert np.array_equal(post[pick].view(np.uint32), posto.view(np.uint32)), (early, "posterior bit patterns")
            assert np.array_equal(V[pick], Vo) and np.array_equal(CWD[pick], cwdo) and np.array_equal(ites[pick], iteso)
        okf = CWD == 1
        assert np.array_equal(V[okf], sent[okf])
        if early:
            assert okf[1::2].mean() > 0.99 and (ites[1::2] < n_ite).mean() > 0.9
        rx.close()


@pytest.mark.parametrize("modcod,F,dbs", [("QPSK-N_8/9", 1100, (3.2, 4.4)), ("QPSK-S_8/9", 2600, (3.2, 4.4)),
                                           ("QPSK-S_3/5", 1300, (0.8, 2.8)), ("32APSK-S_3/4", 1100, (2.4, 3.8))])      # (the last two: the 11- / 13-slot layers with the 16-bit address table)
def test_ldpc_spa_at_size_matches_oracle(O, Rx, modcod, F, dbs):
    """The reference's default decoder (--dec-implem SPA) with more frames than the persistent grid holds: the per-edge message store
    of a workgroup is reused frame after frame (its first-iteration reads are replaced by zeros, like the packed state of the NMS
    kernel).  Two fixed iterations at two Eb/N0 (3.2 / 4.4 dB for rate 8/9): posteriors of the last 6 frames and 6 random ones within the SPA bar of the oracle
    (1e-4 max(1, |L|)); then the converging half with the early stop: hard decisions = the sent word."""
    ch, sent, llr = _big_batch(O, modcod, F, dbs, seed=77, n_cw=4)
    rng = np.random.default_rng(6)
    pick = np.unique(np.concatenate([np.arange(F - 6, F), rng.choice(F - 6, 6, replace=False)]))
    rx = Rx(modcod, max_frames=F, n_ite=2, early_stop=False, implem="SPA")
    V, CWD, post, _ = rx.decode_siho(llr, with_post=True)
    Vo, posto, cwdo, _ = ch.ldpc.decode(llr[pick], n_ite=2, implem=O.SPA_CLIP, sched=O.QC, early_stop=False)
    assert np.all(np.abs(post[pick] - posto) <= 1e-4 * np.maximum(1.0, np.abs(posto))), float(np.abs(post[pick] - posto).max())
    assert (V[pick] != Vo).mean() < 1e-4
    rx.close()
    rx = Rx(modcod, max_frames=F, n_ite=30, early_stop=True, implem="SPA")
    V, CWD = rx.decode_siho(llr)
    assert CWD[1::2].mean() > 0.99 and np.array_equal(V[CWD == 1], sent[CWD == 1])
    rx.close()


@pytest.mark.parametrize("modcod,implem,n_ite", [("QPSK-S_8/9", "NMS", 10), ("QPSK-S_8/9", "SPA", 30), ("QPSK-S_3/5", "SPA", 30), ("QPSK-N_8/9", "NMS", 10)])
def test_ldpc_work_queue_order_changes_no_result(O, Rx, monkeypatch, modcod, implem, n_ite):
    """(round 6) With DVBS2HIP_LDPC_ORDER=1 and the stopping rule the persistent grid's work queue hands out the noisiest frames first (frame_order_launch: sum |LLR| per
    frame, a counting sort; written so that the frames that run to the iteration cap start early -- measured a 0.4 - 5 % loss and therefore opt-in).  Scheduling only: every frame is decoded
    exactly once into its own sockets -- hard decisions, CWD, iteration counts and posteriors of the whole batch are those of the index-order queue (DVBS2HIP_LDPC_ORDER=0),
    bit for bit, on a batch of converging and non-converging frames several times the grid's size; and the order really is by difficulty: the frames that do not converge
    sit in the first part of the queue."""
    F = 2600 if "-S_" in modcod else 1300
    ch, sent, llr = _big_batch(O, modcod, F, (3.4, 4.4) if "8/9" in modcod else (1.0, 2.8), seed=91, n_cw=4)
    rx = Rx(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=True, implem=implem)
    monkeypatch.setenv("DVBS2HIP_LDPC_ORDER", "1")
    a = rx.decode_siho(llr, with_post=True)
    monkeypatch.delenv("DVBS2HIP_LDPC_ORDER")
    b = rx.decode_siho(llr, with_post=True)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    V, CWD, post, ites = a
    assert 0.2 < CWD.mean() < 0.8 and np.array_equal(V[CWD == 1], sent[CWD == 1])
    # the predictor: mean |LLR| separates the two halves of this batch (the test's frames alternate between the two Eb/N0)
    m = np.abs(llr).mean(axis=1)
    hard = np.argsort(m)[:F // 2]
    assert (CWD[hard] == 0).mean() > 0.9
    rx.close()


@pytest.mark.parametrize("modcod,F", [("QPSK-N_8/9", 700), ("QPSK-S_8/9", 1500), ("32APSK-S_3/4", 300)])
@pytest.mark.parametrize("early", [False, True])
def test_plain_decode_siho_equals_the_posterior_socket_form(O, Rx, modcod, F, early):
    """decode_siho with the hard-decision socket only against the call that also asks for the posteriors (other output branches of the
    kernel): same V and CWD for converged and unconverged frames, with and without the stopping rule, more frames than the persistent
    grid.  (Written for an output path that was measured and removed -- docs/negative_results.md -- and kept as a guard of the two socket forms.)"""
    ch, sent, llr = _big_batch(O, modcod, F, (3.0, 4.3) if "QPSK" in modcod else (2.6, 3.6), seed=31, n_cw=4)
    rx = Rx(modcod, max_frames=F, n_ite=6, early_stop=early)
    V1, C1 = rx.decode_siho(llr)
    V2, C2, _, it2 = rx.decode_siho(llr, with_post=True)
    assert np.array_equal(V1, V2) and np.array_equal(C1, C2)
    assert 0 < C1.sum() < F                                   # both kinds of frames were compared
    rx.close()


@pytest.mark.parametrize("implem", ["NMS", "SPA"])
@pytest.mark.parametrize("modcod", ["QPSK-N_8/9", "QPSK-S_8/9", "QPSK-S_3/5", "32APSK-S_3/4"])
def test_ldpc_decisions_are_reproducible_at_size(Rx, modcod, implem):
    """The same batch decoded twice, and its first frames decoded alone, give the same hard decisions, CWD and iteration counts, bit for bit -- on hard frames (nothing
    converges early, every layer of every iteration runs), with every CU holding two workgroups and every workgroup several frames.  A hazard between two instructions of a
    layer loop shows up here as a handful of frames that differ from one call to the next (round 3 met one -- a 16-byte store whose data registers the next
    instruction overwrote -- in a variant of the sum-product layer that passed every three-frame parity test: docs/negative_results.md); the oracle is not needed for this and the batch is the size of a production call."""
    import torch
    dev = torch.device("cuda", 0)
    F = 3072
    torch.manual_seed(77)
    for early, n_ite in ((False, 3), (True, 6)):
        rx = Rx(modcod, max_frames=F, n_ite=n_ite, alpha=1.0, early_stop=early, implem=implem)
        N, K = rx.N_ldpc, rx.K_ldpc
        llr = 2.0 * (1.0 + 0.5 * torch.randn((F, N), device=dev, dtype=torch.float32)) / 0.5 ** 2
        outs = []
        for n in (F, F, 200):
            bits = torch.full((n, K), -1, dtype=torch.int32, device=dev)
            cwd = torch.full((n,), -1, dtype=torch.int8, device=dev)
            torch.cuda.synchronize()                  # the decoder runs on the handle's own stream: the fills have to be over
            rx.decode_siho_dev(llr.data_ptr(), cwd.data_ptr(), bits.data_ptr(), n)
            rx.synchronize()
            outs.append((bits, cwd))
        assert bool((outs[0][0] == outs[1][0]).all()) and bool((outs[0][1] == outs[1][1]).all()), (modcod, implem, early, "two calls differ in %d frames" % int((outs[0][0] != outs[1][0]).any(dim=1).sum()))
        assert bool((outs[0][0][:200] == outs[2][0]).all()) and bool((outs[0][1][:200] == outs[2][1]).all()), (modcod, implem, early, "a subset decodes differently")
        assert int((outs[0][0] < 0).sum()) == 0 and int((outs[0][1] < 0).sum()) == 0          # every socket element written
        rx.close()
