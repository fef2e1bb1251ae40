// a1, production kernel -- layered NMS, one frame per EIGHT-wave workgroup (six waves work), two
// independent workgroups per CU.  Same schedule and arithmetic as the oracle's ORC_SCHED_QC
// (bit-exact), built around what a per-phase cycle profile of its round-1 predecessor showed on MI355X
// (docs/negative_results.md): 30 % of its time was frame I/O issued as ~180 dependent HBM round trips,
// 12 % the replay of same-layer duplicate edges through global memory, and the layer loop itself
// a long per-wave instruction stream (~610 VALU + ~250 scalar per layer; one wave issues one instruction
// every 4.4 cycles whatever its type, tools/probe_valu.hip).
//
//  * Workgroup shape.  The dispatcher deals the waves of a workgroup round-robin over the CU's four
//    SIMDs (tools/probe_placement.hip): 6 waves sit 2+2+1+1, 8 waves 2+2+2+2.  Every wave reads its
//    SIMD from HW_REG_HW_ID and the workgroup learns from a per-CU arrival counter whether it is the
//    first or the second one on its CU (F); the first wave on every SIMD works, the second one only on
//    SIMDs {0,1} (F = 0) or {2,3} (F = 1): the CU's two workgroups load every SIMD with exactly three
//    working waves and share no barrier, so one frame's I/O and memory phases overlap the other
//    frame's arithmetic.  The two idle waves only attend the barriers.  Placement is a performance
//    matter only: any assignment of the six roles decodes correctly.
//  * Frame I/O in batches of 8 independent loads per lane (rows of the posterior image in storage
//    order, LDS rows then global rows); hard decisions for the fused chain are packed with wave
//    ballots instead of 32 strided loads per word.
//  * Static hybrid image (N = 64800): 9 of the 27 slots of every layer are LDS-resident, known at
//    compile time, AND every bit-group with two edges in one layer is among them -- the duplicate-edge
//    replay and the store redirection then never leave LDS, and stores to global memory need no select.
//  * Parity chain in a register (static hybrid): parity bit q t + r joins only checks (r, t) and (r + 1, t) -- the same lane of two
//    consecutive layers.  Its posterior after layer r is read by nobody but layer r + 1, so it travels in a VGPR: slot DEG-2 (p_c)
//    is not stored by layer r < q-1 and slot DEG-1 (p_{c-1}) not loaded by layer r > 0 (layer 0 reads group q-1 shifted by one
//    check: that one goes through memory).  19 loads and 19 stores of the 360 global posterior accesses of an iteration less; the
//    same fp32 values in the same order, so still bit-exact.  The plan keeps the parity groups out of LDS and at the last two slots.
//  * VALU per edge: byte-addressed LDS (one add per access), sign/magnitude merges as single
//    v_and_or / v_bitop3 with the sign mask in an SGPR, store redirection selected on the scalar unit.
#include "dvbs2hip_internal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>

#include "ldpc_det.h"      // w8_det_tanh_half / w8_det_log1p: the tanh-product rule's two functions, correctly rounded operations only (bit for bit the oracle's)
namespace dvbs2 {

typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(4))) uint32_t *const_u32;

constexpr int W8_NL = 9;                    // MODE 3: LDS-resident slots per layer (the first ones)
constexpr int W8_ROW = LDPC_Z * 4;          // bytes per bit-group row
constexpr int W8_IO = 16;                  // independent loads per lane in flight during frame I/O
constexpr uint32_t W8_OOB = 0x7FFFF000u;    // voffset beyond every workspace: the store is dropped
__host__ __device__ __forceinline__ constexpr bool w8_parked(int mode) { return mode == 4 || mode == 5; }      // static hybrid with rows parked in the idle waves' registers
__host__ __device__ __forceinline__ constexpr bool w8_hybrid(int mode) { return mode == 3 || w8_parked(mode); }
// (forced: once the kernel had grown by the round-5 output loops the inliner left this one out of line -- 80 calls per layer, the launch 3.7 x as long)
__host__ __device__ __forceinline__ constexpr bool w8_slot_lds(int mode, int j) { return mode == 0 || (mode == 3 && j < W8_NL) || (w8_parked(mode) && j < ldpc_park_nl(mode)); }

// development knobs of the SPA layer (tools/build_variant.sh): suffix values kept every SPA_BS-th slot; next layer's messages
// requested under the current layer's stores; cache policy of the message traffic
#ifndef SPA_BS
#define SPA_BS 3
#endif
#ifndef SPA_PREFETCH
#define SPA_PREFETCH 1
#endif
#ifndef SPA_AUX
#define SPA_AUX 0
#endif
// Cache policy of the message STORES alone.  Default: non-temporal (aux = 2) where the image does not fit LDS, i.e. for the long codes, whose message
// store (512 x 778 KB for N = 64800) is larger than the Infinity Cache and is read back one whole iteration later -- same-box A/B 64.9 -> 61.1 ms per 16384
// normal frames, 16.6 -> 15.0 ms per 4096; for N = 16200 (512 x 194 KB: cache-resident) the default policy is 7 % faster, and non-temporal LOADS lose on both.
#ifndef SPA_AUX_ST
#define SPA_AUX_ST (MODE == 0 ? SPA_AUX : 2)
#endif
#ifndef NMS_AUX_ST       // cache policy of the min-sum kernel's packed-state stores: 2 (non-temporal) measured 6.95 -> 8.21 ms (the 44 MB of state live in the Infinity Cache)
#define NMS_AUX_ST 0
#endif
#ifndef SPA_MSG4         // 1: a lane keeps the messages of four consecutive slots as one 16-byte piece ([layer][slot / 4][360][4]); 0: [layer][slot][360]
#define SPA_MSG4 1
#endif
#ifndef SPA_ABL          // timing-only ablations (wrong results): 1 no message stores, 2 no message loads, 4 no posterior stores to global memory, 8 no check-node arithmetic
#define SPA_ABL 0
#endif

#ifndef W8_ATAB           // min-sum layer: the slots' addresses come from a per-lane table (L2-resident, requested behind the previous layer's last store) instead of the vector ALU
#define W8_ATAB 1
#endif
#ifndef W8_ABL             // development only (wrong results): what the min-sum layer is sensitive to.  1: pass 2's global stores dropped; 2: pass 1a's global loads replaced by a
#define W8_ABL 0           // register move; 4: pass 1b without the min / sign tracking (3 of 8.5 instructions per slot); 8: pass 2 without the compare and the two selects (3 of 5); 16 / 32: the fused chain's output phase without the packed bytes / the information bits
#endif
#ifndef W8_EARLY_B1        // LDS-only image: the layer's "every read precedes the writes" barrier (duplicate edges) sits right behind pass 1a -- the loads are back within ~100 cycles of the layer's
#define W8_EARLY_B1 1      // start, when the waves have just left the end barrier together -- instead of behind pass 1b, 3000 contended cycles later (QPSK-S 8/9 +2 %, 32APSK-S 3/4 +1.3 %, 3/5 -0.5 %)
#endif
#ifndef W8_P2_GFIRST       // pass 2 on the hybrid images: the global slots first, the LDS slots behind them -- a wave sits out its stores' acknowledgements (s_waitcnt vmcnt(0)) in front of
#define W8_P2_GFIRST 1     // the layer's barrier; issued first they come back under the LDS slots' work (the ablation: no global store in pass 2 is worth 11 % of the launch)
#endif
#ifndef W8_ABS_FOLD       // min-sum layer, pass 1b: min1 = v_min_f32(min1, |x|) with the modifier in the instruction; `fminf(m, fabsf(x))` costs a v_max_f32 |x|, |x| in front of it (IEEE
#define W8_ABS_FOLD 1     // mode: the compiler quiets a signalling NaN first), one instruction of 17 per slot at the 4.25-cycle price (tools/probe_issue4.hip)
#endif
#ifndef W8_SIGN_ADD       // pass 1b: the old message's sign bit moves up by p += p (v_add_u32: 2.1 SIMD cycles) instead of a shift per slot (v_lshlrev_b32: 4.25)
#define W8_SIGN_ADD 1
#endif
#ifndef W8_ATAB_HYB       // the table for the LDS slots of the hybrid images too: measured and left off (docs/negative_results.md: 2 % slower with the request in front of the global stores, 19 % behind them --
                          // the table's loads share the fabric the image's global rows already keep 0.64 busy)
#define W8_ATAB_HYB LDPC_ATAB_HYB      // (dvbs2hip_internal.h: the plan builds the table for the hybrid images only with this set)
#endif
#ifndef W8_DELTA_REG      // LDS-only image: new - old of the duplicate-edge slots kept from the passes instead of rebuilt behind the barrier
#define W8_DELTA_REG 1
#endif
#ifndef W8_FAST_OUT       // output loops of their own for the bits socket alone and for the fused chain (buffer descriptors per frame, scalar descrambling)
#define W8_FAST_OUT 1
#endif
// the fused chain trusts the per-frame BCH flags (dvbs2hip_api.hip: ldpc_verifies_bch), which only the run_out_fast<4|6> output loops write: a build without them would leave
// bch_flag / cwd_bch unwritten while bch_decode_kernel skips frames on those bytes
static_assert(W8_FAST_OUT, "W8_FAST_OUT=0 never writes bch_flag / cwd_bch, which the fused chain's BCH stage reads");
#ifndef W8_SPA_AT16        // sum-product layer, LDS-only image, the 11- and 13-slot codes: the slots' LDS addresses come from a per-lane table of 16-bit entries (two per register) instead of
#define W8_SPA_AT16 LDPC_SPA_AT16      // being formed on the vector ALU (4 instructions per edge, two of them with an SGPR operand); the 27-slot layer has no registers for it (dvbs2hip_internal.h)
#endif
#ifndef W8_SPA_ANYKEY      // sum-product layer: the overflow rule behind a wave-uniform branch per slot (k_ldpc_cu1.hip has it): same-box A/B here QPSK-S 8/9 1697 -> 1654 k,
#define W8_SPA_ANYKEY 0      // 3/5 1178 -> 1151 k frames/s -- 27 scalar branches per layer cut the unrolled slot loop into scheduling regions; off
#endif
#ifndef W8_SPA_TEV        // sum-product layer: the next layer's table in one vector register (what made k_ldpc_cu1.hip's sum-product kernel 12 % faster: there the compiler had parked
#define W8_SPA_TEV 0      // the 32 scalars in vector registers).  Here, same-box A/B: QPSK-S 8/9 1712 -> 1655 k, 3/5 1190 -> 1092 k frames/s, normal frames (mode 4) 315 -> 307 k: off
#endif
#ifndef W8_IN_FAST        // frame input of the information rows, LDS-only image: branch-free batches through a buffer descriptor (see there)
#define W8_IN_FAST 1
#endif
#ifndef W8_OUT_BAR_LGKM   // output phase of the production forms: the barriers wait for the LDS traffic only (same-box A/B: no difference, 6.28 / 6.32 ms under the profiler)
#define W8_OUT_BAR_LGKM 1
#endif
#ifndef W8_OUT_GFIRST     // ... and the rows of the workgroup's global slot are emitted before the LDS rows
#define W8_OUT_GFIRST 1
#endif
#ifndef W8_KEEP_NOWAIT    // the row-keeping waves take a layer's inner barriers without draining their exchanges
#define W8_KEEP_NOWAIT 1
#endif
#ifndef W8_IDX_E32        // pass 2's selects in the 32-bit encoding (inline asm): see the note there.  Measured without effect (round 4: 5.94 against 5.95 ms, same box), off
#define W8_IDX_E32 0
#endif
#ifdef LDPC_PHASE_PROF
#define PROF_MARK_(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof[i] += (uint32_t)(n_ - pt_); pt_ = n_; } while (0)
#ifdef LDPC_PROF_OUT      // the six layer slots show the parts of the OUTPUT phase instead: global rows, LDS rows, the hand-over barriers, parked rows, the wave's fold, the end barrier
#define PROF_MARK(i) do { if ((i) >= 6) PROF_MARK_(i); } while (0)
#define PROF_MARK_O(i) PROF_MARK_(i)
#else
#define PROF_MARK(i) PROF_MARK_(i)
#define PROF_MARK_O(i)
#endif
#else
#define PROF_MARK(i)
#define PROF_MARK_O(i)
#endif

// LDS word at byte offset a of the workgroup's allocation (see the kernel's note on `smem`)
__device__ __forceinline__ lds_float *w8_lds(uint32_t a) { return (lds_float *)(size_t)a; }

// (a & m) | b in one VALU operation (the compiler emits v_and + v_or)
__device__ __forceinline__ float and_or(uint32_t a, uint32_t m_sgpr, float b)
{
    float r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m_sgpr), "v"(b));
    return r;
}

// the lane's index, re-formed where it is needed (two instructions) instead of kept in a register across the layer loop
__device__ __forceinline__ int w8_lane_now()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// message on slot j from the packed per-check state: magnitude c1 at the recorded minimum, c2 elsewhere
// (both >= 0); sign = bit (DEG-1-j) of pk.  `sb` = 0x80000000 held in an SGPR.
template <int DEG>
__device__ __forceinline__ float w8_unpack(float c1, float c2, uint32_t pk, uint32_t j, uint32_t sb)
{
    const float mag = ((pk >> 27) == j) ? c1 : c2;
    return and_or(pk << ((32u - DEG) + j), sb, mag);
}

// exact sum-product pairing (`--dec-implem SPA`, the reference's default), as the oracle's chk_update_spa:
//     a [+] b = sign(a) sign(b) min(|a|,|b|) + log(1 + e^-|a+b|) - log(1 + e^-|a-b|)
// on the hardware exp2 / log2 units.  +inf is the neutral element (absent / NULL slots carry it).
// CHECKED = false: at most one operand is +inf (regular codes: only the absent edge of lane 0 carries it), for which the
// formula itself returns the other operand (e^-inf = 0); CHECKED = true (padded layers, several +inf slots in a row): the
// inf - inf of two neutral elements is caught by selects.
template <bool CHECKED>
__device__ __forceinline__ float w8_boxplus(float a, float b)
{
    const float e1 = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(a + b)), e2 = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(a - b));
    const float l = (__builtin_amdgcn_logf(1.0f + e1) - __builtin_amdgcn_logf(1.0f + e2)) * 0.693147180559945309f;
    const float mn = fminf(fabsf(a), fabsf(b));
    const float sg = __uint_as_float((__float_as_uint(mn) & 0x7FFFFFFFu) | ((__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u));
    const float r = sg + l;
    if (!CHECKED) return r;
    return a == INFINITY ? b : (b == INFINITY ? a : r);
}

typedef float w8_f32x32 __attribute__((ext_vector_type(32)));

// MODE 4 / 5: what the two waves of a workgroup that hold no check do instead of idling -- they keep NR = 32 / 39 bit-group rows of the frame in
// their registers (3 VGPRs per row and lane: lane e of 128 holds elements e, e + 128, e + 256) and swap rows with LDS by the static, cyclic
// schedule of k_ldpc.hip (plan_parked): slot k and one LDS position are shared by a pair of rows whose layers do not interleave; twice per
// iteration, during a layer that uses neither, the slot's row goes to the position and the position's row into the slot.  Every register index
// is a compile-time constant (the loop over the slots is unrolled; the positions come from a table read on the scalar unit).  The waves
// follow the working waves barrier for barrier; `nvote` and the vote words are shared with them.
template <int DEG, int NR>
__device__ __forceinline__ void w8_park_server(const LdpcKParams &p, lds_int *const s_misc, const int wave, const int sidx)
{
    const int q = p.q;
    const const_u32 tab = (const_u32)p.w8.tab, srv = tab + q * LDPC_FAST_STRIDE;
    const const_u32 srow = (const_u32)p.w8.rows + p.w8.nl + p.w8.ng + q;        // bit-group in slot k at the start of an iteration
    // 120 of the 128 lanes hold three elements each: e, e + 120, e + 240 (an exchange must touch every LDS word exactly once)
    const int lane = (int)threadIdx.x & 63, el = sidx * 64 + lane;
    const bool on = el < LDPC_Z / 3;
    const uint32_t a0 = (uint32_t)el * 4u;
    constexpr uint32_t A1 = LDPC_Z / 3 * 4u, A2 = 2u * A1;
    auto lst = [&](uint32_t a, float v) __attribute__((always_inline)) { *w8_lds(a) = v; };
    // one ds_wrxchg_rtn_b32 swaps a register with an LDS word (no temporaries: the rows take 3 NR of the wave's 128 registers)
    auto lxc = [&](uint32_t a, float v) __attribute__((always_inline)) -> float { return __hip_atomic_exchange(w8_lds(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    float R[NR][3];
    int nvote = 0;
    auto vote0 = [&]() __attribute__((always_inline)) -> bool {                    // the working waves' vote (ldpc_wg8_kernel), with nothing to report
        lds_int *const w = s_misc + 10;
        const int k = nvote % 3;
        nvote++;
        if (lane == 0 && wave == 0) w[(k + 1) % 3] = 0;
        __syncthreads();
        return w[k] != 0;
    };
    // (round 4) the swaps of layer r come as a 64-bit mask (bit k: slot k <-> LDS position k; the plan gives pair k position k) behind the [q][NR] byte table: a scalar bit
    // test per slot, every register index and LDS offset a compile-time constant.  The per-slot table entries of round 3 were 39 dependent scalar loads per layer
    // (~3000 cycles of a layer's ~6800), which put these two waves last at the layer's first barrier.
    const const_u32 swm = srv + q * NR;
    auto moves = [&](int r) __attribute__((always_inline)) {
        const uint32_t mlo = swm[2 * r], mhi = swm[2 * r + 1];
        if (on)
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const bool sw = k < 32 ? ((mlo >> k) & 1u) != 0u : ((mhi >> (k - 32)) & 1u) != 0u;
            if (sw) {
                const uint32_t b = (uint32_t)k * (uint32_t)W8_ROW;
                R[k][0] = lxc(b + a0, R[k][0]); R[k][1] = lxc(b + a0 + A1, R[k][1]); R[k][2] = lxc(b + a0 + A2, R[k][2]);
            }
        }
    };
#if W8_KEEP_NOWAIT
    auto bar_inner = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_barrier(); };      // inside a layer: no wait for the exchanges in flight (nobody looks at the rows being swapped before the layer's end barrier)
#else
    auto bar_inner = [&]() __attribute__((always_inline)) { __syncthreads(); };
#endif
    auto layer_barriers = [&](int r) __attribute__((always_inline)) {              // the barriers of one min-sum layer, as the working waves take them
        const const_u32 T = tab + r * LDPC_FAST_STRIDE;
        const uint32_t cinfo = T[28];
        const int ncf = (int)(cinfo & 0xFFu);
        if (ncf > 0) {
            bar_inner();                            // every read of the layer precedes its writes
            bar_inner();                            // the primary writes are in place
            uint32_t prev_lvl = 1u;
            for (int i = (ncf > 1 && ((cinfo >> 21) & 3u) == 1u) ? 2 : 1; i < ncf; i++) {
                const uint32_t lvl = T[48 + i] >> 8;
                if (lvl != prev_lvl) { bar_inner(); prev_lvl = lvl; }
            }
        }
        __syncthreads();                            // end of the layer
    };
    const bool es = p.early_stop != 0;
    // (wave priority 1 / 2 / 3 for these two waves: no difference, 6.00-6.04 ms for every setting, same box)
    for (int qp = blockIdx.x; qp < p.n_frames; ) {
        const int f = p.order ? (int)p.order[qp] : qp;      // (queue position -> frame: LdpcKParams::order)
        const float *Y = p.llr + (size_t)f * p.N;
        const int eln = sidx * 64 + w8_lane_now();  // (re-formed per frame: as a loop invariant the compiler keeps it, widened to 64 bits, across the layer loop -- in registers these waves do not have)
        const int elc = eln < LDPC_Z / 3 ? eln : 0; // (the eight lanes without elements load something harmless)
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uint32_t g = srow[k];
            const float *Yg = Y + (g == 0xFFFFFFFFu ? 0u : g) * (uint32_t)LDPC_Z;
            R[k][0] = __builtin_nontemporal_load(&Yg[elc]); R[k][1] = __builtin_nontemporal_load(&Yg[elc + LDPC_Z / 3]); R[k][2] = __builtin_nontemporal_load(&Yg[elc + 2 * (LDPC_Z / 3)]);
        }
        __syncthreads();                            // the image is in place
        // ph 0: layer r of an iteration; 1: layer r of the syndrome sweep; 2: catching up with the schedule after a sweep that stopped early
        int ph = 0, r = 0, it = 0;
        bool ok = false, fin = false;
        while (!fin) {
            bool mv = true;
            if (ph == 1 && es && r == 0 && vote0()) { ok = false; mv = false; ph = 3; }       // stopping rule: the vote on layer 0 comes before anything moves
            if (mv) moves(r);
            if (ph == 0) {
                layer_barriers(r);
                if (++r == q) { r = 0; it++; if (es || it == p.n_ite) { ph = 1; ok = true; } }
            } else if (ph == 1) {
                bool bad = false;
                if (es) { if (r == 0) __syncthreads(); else bad = vote0(); }
                else { if (r == q - 1) { if (vote0()) ok = false; } else __syncthreads(); }
                if (bad) { ok = false; ph = 2; }
                if (++r == q) { if (ph == 2) __syncthreads(); ph = 3; }
            } else if (ph == 2) {
                if (++r == q) { __syncthreads(); ph = 3; }
            }
            if (ph == 3) { if (ok || it >= p.n_ite) fin = true; else { ph = 0; r = 0; } }
        }
        // outputs: the parked rows go through LDS positions 0 .. NR-1 once the working waves have read the LDS rows
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NR; k++) if (on) { const uint32_t b = (uint32_t)k * (uint32_t)W8_ROW; lst(b + a0, R[k][0]); lst(b + a0 + A1, R[k][1]); lst(b + a0 + A2, R[k][2]); }
        __syncthreads();
        __syncthreads();                            // the image is reused by the next frame
        qp = s_misc[9];
    }
}

// SPA = 0: normalised min-sum with the packed per-check state.  SPA = 1: sum-product check node (exact, complement-product domain), the c->v
// messages kept per edge (fp32, [layer][slot][360] after the image in the workgroup's global slot).  SPA = 2: the same layer with the check node as
// AFF3CT's Update_rule_SPA evaluates it (tanh product in fp32, saturating), bit for bit the oracle's ORC_SPA_TANH (w8_det_*).  SPA = 3: `--dec-implem SPA`, the exact node with
// every message clipped at AFF3CT's cap -- which makes the scale, the running minima and the overflow rule of SPA = 1 unnecessary.
template <int DEG, int MODE, int SPA = 0>      // MODE 0: image in LDS, 1: in the workgroup's global slot, 3: static hybrid
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((SPA == 2 && DEG > 13) ? 2 : 4, 4)))      // (the tanh-product rule on 27 slots: one workgroup per CU, 256 registers -- a parity mode, not a throughput one)
ldpc_wg8_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    // LDS is addressed by plain byte offsets (w8_lds): the kernel has no static LDS, so its dynamic block starts at 0.  Through the symbol every address the
    // layer loop keeps in a register would carry an "+ smem" that the compiler resolves to an add of 0 only at link time -- one vector instruction per access.
    if ((uint32_t)(size_t)(lds_float *)smem != 0u) __builtin_trap();
    lds_int *const s_misc = (lds_int *)w8_lds((uint32_t)p.w8.lds_bytes - (uint32_t)LDPC_W8_MISC_BYTES);      // [0..7] SIMD of wave w, [8] F, .. (LDPC_W8_MISC_BYTES)
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane0 = (int)threadIdx.x & 63;
    const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);            // HW_REG_HW_ID
    const int simd = (int)((hw >> 4) & 3u);
    if (lane0 == 0) s_misc[wave] = simd;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);        // HW_REG_XCC_ID
        const uint32_t key = (xcc & 15u) << 8 | ((hw >> 13) & 7u) << 5 | ((hw >> 12) & 1u) << 4 | ((hw >> 8) & 15u);
        s_misc[8] = p.cu_ctr ? (int)(atomicAdd(&p.cu_ctr[key], 1u) & 1u) : 0;
        s_misc[10] = 0; s_misc[11] = 0; s_misc[12] = 0;      // SPA: the three vote words
        for (int k = 16; k < 22; k++) s_misc[k] = 0;         // fused chain: the frame's BCH remainder (zeroed again by the thread that reads it)
    }
    int nvote = 0;                                           // SPA: votes taken so far (rotation of the vote words)
    __syncthreads();
    int role;
    {
        int k = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int w = 0; w < 8; w++) {
            const int s = s_misc[w];
            c0 += s == 0; c1 += s == 1; c2 += s == 2; c3 += s == 3;
            if (w < wave && s == simd) k++;
        }
        const bool balanced = c0 == 2 && c1 == 2 && c2 == 2 && c3 == 2;
        if (!balanced) role = wave < 6 ? wave : -1;
        else if (k == 0) role = simd;
        else role = ((simd >> 1) == s_misc[8]) ? 4 + (simd & 1) : -1;
        role = __builtin_amdgcn_readfirstlane(role);
        if (w8_parked(MODE) && role < 0) {               // the two waves without checks keep parked rows (w8_park_server): which of the two is this one?
            int sidx = 0;
            for (int w = 0; w < wave; w++) {
                const int s = s_misc[w];
                int kw = 0;
                for (int w2 = 0; w2 < w; w2++) kw += s_misc[w2] == s;
                const int rw = !balanced ? (w < 6 ? w : -1) : kw == 0 ? s : ((s >> 1) == s_misc[8] ? 4 + (s & 1) : -1);
                sidx += rw < 0;
            }
            w8_park_server<DEG, ldpc_park_nr(MODE)>(p, s_misc, wave, __builtin_amdgcn_readfirstlane(sidx));
#ifdef LDPC_PHASE_PROF
            if (lane0 == 0 && p.cu_ctr) for (int i = 0; i < 12; i++) p.cu_ctr[LDPC_CU_CTR_WORDS + ((int)blockIdx.x * 8 + wave) * 12 + i] = 0u;
#endif
            return;
        }
    }
    const int lane = w8_lane_now() & 63;      // (re-formed behind the role split: the row-keeping waves' branch reuses the register, and the one from the kernel's start was spilled around it)
    const int t = role * 64 + lane;
    const bool act = role >= 0 && t < LDPC_Z;
    const uint32_t t4 = (uint32_t)t * 4u;
    const int q = p.q, M = p.M;
    float *gwork = p.gwork + (size_t)blockIdx.x * p.gwork_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gwork, 0, p.gwork_words * 4, 0x00020000);
    const const_u32 tab = (const_u32)p.w8.tab;
    const const_u32 rows = (const_u32)p.w8.rows;
    const uint32_t st_base = p.w8.st_base;                   // packed state: [q][360] x {c1, c2, pk} (12 bytes per check, one access)
    const uint32_t ljunk = p.w8.lds_junk;                    // LDS junk row (write-only)
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));                             // the sign mask as an SGPR operand (VOP3 takes no literal)
    // Offset of a buffer store of MORE THAN 8 BYTES: all of it in the vector register, the scalar offset field zero.  With a scalar offset register the compiler assumes that
    // the next instruction may overwrite the store's data registers at once (the rule of the first GCN parts); on gfx950 that corrupts the stored data -- found when a variant of
    // the sum-product layer without branches behind its 16-byte message stores decoded differently from call to call.  With the field zero the compiler's hazard recognizer
    // puts the wait state in itself (one vector add per store is the price; the opaque copy keeps instruction selection from moving the scalar part back).
    auto wide_off = [&](uint32_t voff, uint32_t soff) __attribute__((always_inline)) -> uint32_t { uint32_t o = voff + soff; asm volatile("" : "+v"(o)); return o; };
    auto gld = [&](uint32_t voff, uint32_t soff) __attribute__((always_inline)) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0)); };
    auto gst = [&](uint32_t voff, uint32_t soff, float v) __attribute__((always_inline)) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, voff, soff, 0); };
    auto gld1 = [&](uint32_t voff, uint32_t soff) __attribute__((always_inline)) { if (W8_ABL & 2) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(voff | 0x3F000000u)); return r; } return gld(voff, soff); };
    auto gst2 = [&](uint32_t voff, uint32_t soff, float v) __attribute__((always_inline)) { if (W8_ABL & 1) { asm volatile("" :: "v"(v), "v"(voff), "s"(soff)); return; } gst(voff, soff, v); };
    auto mld = [&](uint32_t voff, uint32_t soff) __attribute__((always_inline)) { if (SPA_ABL & 2) return __uint_as_float(voff & 0x3F000000u); return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, SPA_AUX)); };      // SPA messages
    auto mst = [&](uint32_t voff, uint32_t soff, float v) __attribute__((always_inline)) { if (SPA_ABL & 1) { asm volatile("" :: "v"(v)); return; } __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, voff, soff, SPA_AUX); };
    // SPA_MSG4: the messages of a layer as [slot / 4][360][4 slots] (the last group holds DEG mod 4 of them): a lane's messages of four consecutive slots are 16
    // consecutive bytes, a wave's access 1 KB -- whole lines written and read by ONE instruction instead of four 256-byte pieces of four rows
    // (the per-edge message stream, HBM by construction, is what the sum-product kernel is bound by: docs/negative_results.md's ablations).  Same bytes per layer.
    typedef uint32_t m_u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t m_u32x3 __attribute__((ext_vector_type(3)));
    typedef uint32_t m_u32x2 __attribute__((ext_vector_type(2)));
    constexpr int MG4 = DEG / 4, MR = DEG % 4;                                            // whole groups of four slots, slots in the last group
    auto mgrp_ld = [&](float *dst, uint32_t tb, uint32_t lbase) __attribute__((always_inline)) {                         // all DEG messages of this lane; tb = 4 t (bytes of one dword per lane)
#pragma unroll
        for (int g = 0; g < MG4; g++) {
            m_u32x4 v = {0u, 0u, 0u, 0u};
            if (!(SPA_ABL & 2)) v = __builtin_amdgcn_raw_buffer_load_b128(rs, tb * 4u, lbase + (uint32_t)g * (W8_ROW * 4u), SPA_AUX);
            dst[4 * g] = __uint_as_float(v.x); dst[4 * g + 1] = __uint_as_float(v.y); dst[4 * g + 2] = __uint_as_float(v.z); dst[4 * g + 3] = __uint_as_float(v.w);
        }
        const uint32_t tbase = lbase + (uint32_t)MG4 * (W8_ROW * 4u);
        if (MR == 3) { m_u32x3 v = {0u, 0u, 0u}; if (!(SPA_ABL & 2)) v = __builtin_amdgcn_raw_buffer_load_b96(rs, tb * 3u, tbase, SPA_AUX);
                       dst[4 * MG4] = __uint_as_float(v.x); dst[4 * MG4 + 1] = __uint_as_float(v.y); dst[4 * MG4 + 2] = __uint_as_float(v.z); }
        if (MR == 2) { m_u32x2 v = {0u, 0u}; if (!(SPA_ABL & 2)) v = __builtin_amdgcn_raw_buffer_load_b64(rs, tb * 2u, tbase, SPA_AUX);
                       dst[4 * MG4] = __uint_as_float(v.x); dst[4 * MG4 + 1] = __uint_as_float(v.y); }
        if (MR == 1) dst[4 * MG4] = mld(tb, tbase);
    };
    auto mone_ld = [&](uint32_t slot, uint32_t tb, uint32_t lbase) __attribute__((always_inline)) -> float {             // one message (wave-uniform slot): the duplicate edges' old values
        const uint32_t g = slot >> 2, k = slot & 3u;
        return g < (uint32_t)MG4 ? mld(tb * 4u, lbase + g * (W8_ROW * 4u) + k * 4u) : mld(tb * (uint32_t)(MR ? MR : 1), lbase + (uint32_t)MG4 * (W8_ROW * 4u) + k * 4u);
    };
    auto lld = [&](uint32_t a) __attribute__((always_inline)) -> float { return *w8_lds(a); };
    auto lst = [&](uint32_t a, float v) __attribute__((always_inline)) { *w8_lds(a) = v; };
    const int nl_info = p.w8.nl_info, nl = p.w8.nl, ng_info = p.w8.ng_info, ng = p.w8.ng;
    const uint32_t grow0 = 2u * W8_ROW;                      // global image: [junk row][+inf row][group rows ..]
    // (LDS-only image only: on the hybrid image of the normal frames the table's loads queue behind the global rows' and the kernel is 19 % SLOWER, 6.73 against 5.65 ms)
    constexpr bool ATAB = W8_ATAB && !SPA && (MODE == 0 || (W8_ATAB_HYB && w8_hybrid(MODE)));
    constexpr bool AT16 = W8_SPA_AT16 && SPA && MODE == 0 && DEG <= LDPC_SPA_AT16_MAXDEG;      // (the plan builds the table for exactly this case: k_ldpc.hip, LDPC_SPA_AT16)
    constexpr int ND16 = (DEG + 1) / 2, NP16 = (ND16 + 3) / 4;
    const __amdgpu_buffer_rsrc_t rs_at = __builtin_amdgcn_make_buffer_rsrc((void *)((ATAB || AT16) ? p.w8.atab : (const uint32_t *)p.w8.tab), 0, ATAB ? p.q * ((DEG + 3) / 4) * (LDPC_AT_LANES * 16) : AT16 ? p.q * NP16 * (LDPC_AT_LANES * 16) : 0, 0x00020000);
    const uint32_t at_vo = (uint32_t)(role >= 0 ? t : 0) * 16u;
    constexpr bool FWD = w8_hybrid(MODE);                    // parity chain forwarded in a register (plan: p_c at slot DEG-2, p_{c-1} at DEG-1, both global)
#ifdef LDPC_PHASE_PROF
    uint32_t prof[12];
    for (int i = 0; i < 12; i++) prof[i] = 0u;
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long pt_ = prof_t0;
#endif

    // Frames are handed out through a counter (the first one of every workgroup is its block index): with the syndrome early
    // stop frames take 1 .. n_ite iterations, and a fixed round-robin assignment left the workgroups with the slow frames running
    // alone at the end of the launch.
    // (round 6) The queue can hand the frames out in an order of its own (LdpcKParams::order: the noisiest first, frame_order_launch in k_ldpc.hip) -- opt-in, measured a loss.
    for (int qp = blockIdx.x; qp < p.n_frames; ) {
        const int f = p.order ? (int)__builtin_amdgcn_readfirstlane((int)p.order[qp]) : qp;
        // ---- channel LLRs -> posterior image, W8_IO independent loads per lane in flight; packed state := 0
        const float *Y = p.llr + (size_t)f * p.N;
        // the frame I/O addresses are formed per frame from an opaque copy of the lane's index: hoisted out of the frame loop they are
        // ~10 64-bit pointers per lane that the layer loop has no registers for (they were spilled)
        // (same-box A/B of the branch-free input form below: LDS-only image 4.43 -> 4.40 ms per 16384 short frames, hybrid image 5.755 -> 5.81 ms per 4096 normal frames -- its
        // input phase is 14 % shorter there too, and the launch longer: the burst of 16 back-to-back loads lands on the fabric the CU's other workgroup is decoding through)
        constexpr bool INF = W8_IN_FAST && MODE == 0;
        const int lio = (SPA || INF) ? w8_lane_now() : lane, tio = (SPA || INF) ? role * 64 + lio : t;
        const uint32_t tio4 = (SPA || INF) ? (uint32_t)tio * 4u : t4;
        __builtin_amdgcn_s_setprio(3);                       // frame I/O: short bursts of loads that the other workgroup's arithmetic should not delay
        if (INF && act) {
            // info groups: coalesced rows of 360, streamed in once (non-temporal).  (round 5) Batches without a branch or a clamp in them -- the segment's remainder is its last
            // W8_IO rows once more, copied twice -- through a buffer descriptor of the frame: the row numbers of a batch arrive as one wide scalar load and a row's load is ONE
            // instruction (row offset in the scalar field).  With a clamped index per row the compiler fetched every row number by itself and waited for it in front of the
            // row's load (16 scalar round trips per batch between loads that are meant to go out back to back), and formed a 64-bit address per lane and row.
            const int fin = __builtin_amdgcn_readfirstlane(f);
            const __amdgpu_buffer_rsrc_t rs_llr = __builtin_amdgcn_make_buffer_rsrc((void *)(p.llr + (size_t)fin * p.N), 0, p.N * 4, 0x00020000);
            auto in_body = [&](int rb, int l0, bool lds, auto b_c) __attribute__((always_inline)) {
                constexpr int B = decltype(b_c)::value;
                float v[B];
#pragma unroll
                for (int k = 0; k < B; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_llr, tio4, rows[rb + l0 + k] * (uint32_t)W8_ROW, 2));
#pragma unroll
                for (int k = 0; k < B; k++) { if (lds) lst((uint32_t)(l0 + k) * W8_ROW + tio4, v[k]); else gst(tio4, grow0 + (uint32_t)(l0 + k) * W8_ROW, v[k]); }
            };
            auto in_seg = [&](int rb, int n, bool lds) __attribute__((always_inline)) {
                int l0 = 0;
                for (; l0 + W8_IO <= n; l0 += W8_IO) in_body(rb, l0, lds, std::integral_constant<int, W8_IO>{});
                if (l0 < n) {
                    if (n >= W8_IO) in_body(rb, n - W8_IO, lds, std::integral_constant<int, W8_IO>{});
                    else for (; l0 < n; l0++) in_body(rb, l0, lds, std::integral_constant<int, 1>{});
                }
            };
            in_seg(0, nl_info, true);
            in_seg(nl, ng_info, false);
        }
        if (!INF && act) {
            // info groups: coalesced rows of 360, streamed in once (non-temporal)
            for (int l0 = 0; l0 < nl_info; l0 += W8_IO) {
                float v[W8_IO];
#pragma unroll
                for (int k = 0; k < W8_IO; k++) v[k] = __builtin_nontemporal_load(&Y[(int)rows[l0 + k < nl_info ? l0 + k : nl_info - 1] * LDPC_Z + tio]);
#pragma unroll
                for (int k = 0; k < W8_IO; k++) if (l0 + k < nl_info) lst((uint32_t)(l0 + k) * W8_ROW + tio4, v[k]);
            }
            for (int l0 = 0; l0 < ng_info; l0 += W8_IO) {
                float v[W8_IO];
#pragma unroll
                for (int k = 0; k < W8_IO; k++) v[k] = __builtin_nontemporal_load(&Y[(int)rows[nl + (l0 + k < ng_info ? l0 + k : ng_info - 1)] * LDPC_Z + tio]);
#pragma unroll
                for (int k = 0; k < W8_IO; k++) if (l0 + k < ng_info) gst(tio4, grow0 + (uint32_t)(l0 + k) * W8_ROW, v[k]);
            }
        }
        if (role >= 0) {
            // parity part: bit K + q t + r belongs to row n_info + r, element t -- a stride-q gather if read row by row (64
            // cache lines per wave load).  Each wave instead reads the q * 64 consecutive LLRs of its 64 checks coalesced and
            // scatters them into the rows (LDS, or the workgroup's cache-resident global slot).
            const int tl0 = role * 64, cnt = (LDPC_Z - tl0 < 64 ? LDPC_Z - tl0 : 64) * q;
            const float *Yp = Y + p.K + tl0 * q;
            const const_u32 prow = rows + nl + ng;
            // element e = 64 i + lane of the wave's region is (check tl = e / q, parity group r = e mod q): stepped, not divided
            const int dq = 64 / q, dr = 64 - dq * q;
            int tl = lio / q, r = lio - tl * q;
            constexpr int PIO = 8;
            for (int i0 = 0; i0 < q; i0 += PIO) {
                float v[PIO];
#pragma unroll
                for (int k = 0; k < PIO; k++) { const int e = (i0 + k) * 64 + lio; v[k] = e < cnt ? Yp[e] : 0.f; }
#pragma unroll
                for (int k = 0; k < PIO; k++) {
                    if ((i0 + k) * 64 + lio < cnt) {
                        const uint32_t loc = prow[r], off = (loc & 0x7FFFFFFFu) + (uint32_t)(tl0 + tl) * 4u;
                        if (loc >> 31) gst(off, 0u, v[k]); else lst(off, v[k]);
                    }
                    tl += dq; r += dr;
                    if (r >= q) { r -= q; tl++; }
                }
            }
        }
        if (act) {
            // the c->v state is NOT zeroed: a layer's state is first read one iteration after it was first written, and the
            // reads of the first iteration are replaced by zeros where they happen (nx* / dl[] below)
            if (p.w8.pad) { if (MODE == 0) lst(ljunk + W8_ROW + t4, INFINITY); else gst(t4, W8_ROW, INFINITY); }     // what NULL slots read
            if (AT16 && t == 0) lst(ljunk + 4u, INFINITY);      // ... through the 16-bit address table: one word inside the first 64 KB (the junk row is written at its word 0 only)
        }
        if (p.packed && (SPA ? wave == 0 && lio == 0 : threadIdx.x == 0) && (p.K & 31)) p.packed[(size_t)f * ((p.K + 31) / 32) + p.K / 32] = 0u;     // the bits behind K in the last word
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        PROF_MARK(8);

        int it = 0;
        bool ok = false;
        float nx1 = 0.f, nx2 = 0.f, nxk = 0.f;           // packed state of the next layer (prefetched)
        float pfw = 0.f;                                 // posterior of parity bit q t + r after layer r, on its way to layer r + 1
        float onx[SPA ? DEG : 1];                        // SPA: old messages of the NEXT layer, requested while this layer's stores drain
#pragma unroll
        for (int j = 0; j < (SPA ? DEG : 1); j++) onx[j] = 0.f;
        // layer table of the NEXT layer, fetched under the end-of-layer barrier: 27 slots | prim | conflict info | 2 conflict entries
        uint32_t TE[32];
        constexpr bool TEV = SPA && W8_SPA_TEV;      // sum-product kernel: the next layer's table in ONE vector register (lane j = entry j, read back with v_readlane) instead of 32 scalars carried across the layer's barriers (k_ldpc_cu1.hip: the compiler parked them in vector registers)
#pragma unroll
        for (int j = 0; j < 32; j++) TE[j] = TEV ? 0u : tab[j];
        uint32_t tev = TEV ? p.w8.tab[w8_lane_now() & 31] : 0u;
        // (round 4) the slots' addresses of the NEXT layer, 16 bytes (four slots) per load from the per-lane table
        constexpr int NW4 = (DEG + 3) / 4;
        uint32_t w[4 * NW4];                        // (loop-carried: the addresses of the current layer until its stores have been issued, then the next layer's)
        constexpr int AT_NS = MODE == 0 ? DEG : (MODE == 3 ? W8_NL : w8_parked(MODE) ? ldpc_park_nl(MODE) : 0);      // slots whose address the table supplies (hybrid: the LDS slots, the first ones)
        auto at_request = [&](int rl) __attribute__((always_inline)) {
            typedef uint32_t at_u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t at_u32x3 __attribute__((ext_vector_type(3)));
            typedef uint32_t at_u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int g4 = 0; g4 < (AT_NS + 3) / 4; g4++) {
                const uint32_t so = (uint32_t)((rl * NW4 + g4) * (LDPC_AT_LANES * 16));
                const int left = AT_NS - 4 * g4;          // (the last piece loads only what the table supplies: the registers behind it hold global slots' offsets)
                if (left >= 4 || MODE == 0) { const at_u32x4 q4 = __builtin_amdgcn_raw_buffer_load_b128(rs_at, at_vo, so, 0); w[4 * g4] = q4.x; w[4 * g4 + 1] = q4.y; w[4 * g4 + 2] = q4.z; w[4 * g4 + 3] = q4.w; }
                else if (left == 3) { const at_u32x3 q3 = __builtin_amdgcn_raw_buffer_load_b96(rs_at, at_vo, so, 0); w[4 * g4] = q3.x; w[4 * g4 + 1] = q3.y; w[4 * g4 + 2] = q3.z; }
                else if (left == 2) { const at_u32x2 q2 = __builtin_amdgcn_raw_buffer_load_b64(rs_at, at_vo, so, 0); w[4 * g4] = q2.x; w[4 * g4 + 1] = q2.y; }
                else w[4 * g4] = __builtin_amdgcn_raw_buffer_load_b32(rs_at, at_vo, so, 0);
            }
        };
#pragma unroll
        for (int j = 0; j < 4 * NW4; j++) w[j] = 0u;
        if (ATAB && role >= 0) at_request(0);
        uint32_t w16[AT16 ? 4 * NP16 : 1];          // (sum-product, LDS-only image) the 16-bit addresses of the current layer's slots, two per register; the next layer's once this one's stores are out
        auto at16_request = [&](int rl) __attribute__((always_inline)) {
            typedef uint32_t at_u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t at_u32x3 __attribute__((ext_vector_type(3)));
            typedef uint32_t at_u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int g4 = 0; g4 < NP16; g4++) {
                const uint32_t so = (uint32_t)((rl * NP16 + g4) * (LDPC_AT_LANES * 16));
                const int left = ND16 - 4 * g4;
                if (left >= 4) { const at_u32x4 q4 = __builtin_amdgcn_raw_buffer_load_b128(rs_at, at_vo, so, 0); w16[4 * g4] = q4.x; w16[4 * g4 + 1] = q4.y; w16[4 * g4 + 2] = q4.z; w16[4 * g4 + 3] = q4.w; }
                else if (left == 3) { const at_u32x3 q3 = __builtin_amdgcn_raw_buffer_load_b96(rs_at, at_vo, so, 0); w16[4 * g4] = q3.x; w16[4 * g4 + 1] = q3.y; w16[4 * g4 + 2] = q3.z; }
                else if (left == 2) { const at_u32x2 q2 = __builtin_amdgcn_raw_buffer_load_b64(rs_at, at_vo, so, 0); w16[4 * g4] = q2.x; w16[4 * g4 + 1] = q2.y; }
                else w16[4 * g4] = __builtin_amdgcn_raw_buffer_load_b32(rs_at, at_vo, so, 0);
            }
        };
#pragma unroll
        for (int j = 0; j < (AT16 ? 4 * NP16 : 1); j++) w16[j] = 0u;
        if (AT16 && role >= 0) at16_request(0);
        while (it < p.n_ite) {
            for (int r = 0; r < q; r++) {
                const const_u32 T = tab + r * LDPC_FAST_STRIDE;
                uint32_t E[DEG];
#pragma unroll
                for (int j = 0; j < DEG; j++) E[j] = TEV ? (uint32_t)__builtin_amdgcn_readlane((int)tev, j) : TE[j];
                const uint32_t prim = TEV ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 27) : TE[27], cinfo = TEV ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 28) : TE[28],
                               ce0 = TEV ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 29) : TE[29], ce1 = TEV ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 30) : TE[30];
                const int ncf = (int)(cinfo & 0xFFu);
                const bool mask0 = (r == 0) && (t == 0);        // p_{c-1} of check 0 does not exist
                if (SPA) {
                    // ================= sum-product layer =================
                    // Exact check node in the COMPLEMENT-PRODUCT domain (what the oracle's pairwise boxplus recursions evaluate):
                    //     |out_j| = 2 atanh( prod_{i != j} tanh(a_i / 2) ),  a_i = |v->c_i|.
                    // With u_i = 1 - tanh(a_i / 2) = 2 e^-a_i / (1 + e^-a_i) -- precise however large a_i is, where tanh itself saturates --
                    // Q_j = 1 - prod_{i != j} (1 - u_i) follows from prefix and suffix recursions Q_ab = Q_a + Q_b (1 - Q_a) made of
                    // additions of positive terms only (no cancellation, unlike prod / tanh(a_j / 2)), and |out_j| = ln((2 - Q_j) / Q_j).
                    // To keep Q inside the fp32 range for LLRs of hundreds, the lane carries Q' = 2^s2 Q with s2 = max(0, (min2 - 16) log2 e):
                    // an exact change of variable (Q'_ab = Q'_a + Q'_b (1 - kap Q'_a), kap = 2^-s2), chosen from the second smallest
                    // magnitude so that the sum seen by the WEAKEST edge is representable; should the weakest edge itself then overflow
                    // (min2 - min1 > 60), the other edges' outputs are min1 to within e^-57.  Per edge: 1 exp + 1 rcp on the way in, 1 rcp + 1 log
                    // on the way out and ~20 full-rate operations, against 2.8 boxplus x (2 exp + 2 log + 12) of the forward / backward form.
                    constexpr uint32_t mpitch = W8_ROW;      // message rows packed like the image's (a run-time pitch costs two scalar instructions per access; 1536-byte rows -- whole lines -- measured SLOWER, docs/negative_results.md)
                    const uint32_t mrow = st_base + (uint32_t)(r * DEG) * mpitch;       // messages of this layer: [slot][360 of mpitch / 4]
                    const uint32_t dupmask = TEV ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 31) : TE[31];
                    // the circulant offsets are formed twice, for the loads and again for the stores (an opaque copy of t4 keeps the compiler
                    // from holding 27 of them across the arithmetic: registers, not instructions, are what this layer is short of)
                    uint32_t t4s = t4, MAGM = 0x7FFFFFFFu;
                    asm volatile("" : "+s"(MAGM));                // the magnitude mask as an SGPR operand (VOP3 takes no literal)
                    auto woff = [&](int j) __attribute__((always_inline)) { const uint32_t d = t4 - (E[j] & 0x7FFu); return min(d, d + (uint32_t)W8_ROW); };
                    auto woff_s = [&](int j) __attribute__((always_inline)) { const uint32_t d = t4s - (E[j] & 0x7FFu); return min(d, d + (uint32_t)W8_ROW); };
                    // (round 5, LDS-only image) ... or read from the per-lane table: slot j's LDS address is one half of a register (a v_and with 0xFFFF or a shift by 16)
                    uint32_t M16 = 0xFFFFu;
                    asm volatile("" : "+v"(M16));                 // in a vector register: the plain two-register v_and_b32 issues every 2.07 cycles, with a literal 2.6, with an SGPR 4.25
                    auto a16 = [&](int j) __attribute__((always_inline)) -> uint32_t { const uint32_t wv = w16[AT16 ? j / 2 : 0]; return (j & 1) ? wv >> 16 : wv & M16; };
                    constexpr int KDS = ldpc_w8_kd(DEG);          // (plan contract, LDS-only image: from this slot on every slot is a primary edge)
                    auto dup_slot = [&](int i) __attribute__((always_inline)) -> uint32_t { return i == 0 ? (cinfo >> 8) & 31u : i == 1 ? (cinfo >> 16) & 31u : T[48 + i] & 31u; };
                    // suffix values are kept for every BS-th slot only and rebuilt from there on the way forward (one or two steps off the
                    // critical path): the full array does not fit the 128-VGPR budget of two workgroups per CU beside x[] and u[]
                    constexpr int BS = DEG > 13 ? SPA_BS : 1, NB = (DEG + BS - 1) / BS;      // (the 11- and 13-slot codes have the registers for every suffix value)
                    float x[DEG], u[DEG], B[NB];                    // v->c ; 2^s2 (1 - tanh(|v->c| / 2)) ; suffix recursion
                    // The plan puts the duplicate edges into the first slots of a layer, conflict entry i = slot i, so what such an edge adds to its
                    // bit in the replay, new - old message, is ONE subtraction per slot against the old message that came in with the others: no extra loads, no per-slot test
                    // and six-way match against the conflict list (QPSK-S 8/9: 11.05 -> 9.9 ms per 16384 frames).  (This form first decoded differently from call to call: with
                    // the branches of the match gone, nothing stood between a 16-byte message store and the next vector write to its data registers -- see wide_off.)
                    constexpr bool OD_STATIC = true;          // (every image mode: the plan orders the slots of a sum-product plan this way in all of them)
                    float od[LDPC_SPA_MAXC];                        // old c->v of the duplicate edges, then new - old (what such an edge adds)
                    float mn1 = INFINITY, kap = 1.f, cln = 0.f, key = 0.f;
                    uint32_t sx = 0u;
                    w8_f32x32 tv;                                    // SPA = 2: tanh(|v->c| / 2) per slot (a vector: read by wave-uniform index for the ordered product)
                    float tprod = 1.f;
                    auto comb = [&](float a, float b) __attribute__((always_inline)) { return __builtin_fmaf(b, __builtin_fmaf(-kap, a, 1.f), a); };      // Q'_ab
#pragma unroll
                    for (int i = 0; i < LDPC_SPA_MAXC; i++) od[i] = 0.f;
                    if (act) {
                        __builtin_amdgcn_s_setprio(3);            // as in the min-sum layer: load issue first, the long arithmetic last ((3, 1) / (2, 1) in and behind pass 2 instead of (2, 0): within 0.4 %)
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            if (AT16) { x[j] = lld(a16(j)); continue; }
                            const uint32_t base = (E[j] >> 11) & 0x3FFFFu, wj = woff(j);
                            if (FWD && j == DEG - 1 && r > 0) x[j] = pfw;                  // p_{c-1}: handed over by layer r - 1
                            else x[j] = w8_slot_lds(MODE, j) ? lld(wj + base) : gld(wj, base);
                        }
                        if (it > 0) {
#if !SPA_PREFETCH
#if SPA_MSG4
                            mgrp_ld(onx, t4, mrow);
#else
#pragma unroll
                            for (int j = 0; j < DEG; j++) onx[j] = mld(t4, mrow + (uint32_t)j * mpitch);     // old message
#endif
#endif
                            if (!OD_STATIC) {
#pragma unroll
                                for (int i = 0; i < LDPC_SPA_MAXC; i++) if (i < ncf) od[i] = SPA_MSG4 ? mone_ld(dup_slot(i), t4, mrow) : mld(t4, mrow + dup_slot(i) * mpitch);
                            }
                        }
                        __builtin_amdgcn_s_setprio(0);
                        if (OD_STATIC) {
#pragma unroll
                            for (int j = 0; j < LDPC_SPA_MAXC && j < DEG; j++) od[j] = onx[j];
                        }
#pragma unroll
                        for (int j = 0; j < DEG; j++) x[j] = x[j] - onx[j];      // zeros in the first iteration
                        if (mask0) x[DEG - 1] = INFINITY;
                        if constexpr (SPA == 2) {
                            // the slots in the oracle's edge order, 5 bits each (k_ldpc.hip): wave-uniform indices into the lane's tanh values
                            uint32_t pw[5];
#pragma unroll
                            for (int k = 0; k < 5; k++) pw[k] = T[LDPC_TANH_ORDER + k];
#pragma unroll
                            for (int j = 0; j < DEG; j++) { sx ^= __float_as_uint(x[j]); tv[j] = w8_det_tanh_half(fabsf(x[j])); }
#pragma unroll
                            for (int c = 0; c < DEG; c++) tprod = tprod * tv[(pw[c / 6] >> (5 * (c % 6))) & 31u];
                        } else if constexpr (SPA == 3) {
                            // (round 6) `--dec-implem SPA`: every message is clipped at 16.64 on the way out, i.e. wherever Q = 1 - prod (1 - u_i) falls below 1.2e-7 the clip
                            // decides -- Q needs no more range than plain fp32 gives (u_i = 2 / (e^a_i + 1) may underflow to 0: a sum of zeros is below 1.2e-7 too, 2 / 0 = +inf,
                            // log = +inf, the clip takes it), so the per-check scale 2^s2 of the unclipped rule, the two running minima it is chosen from and the
                            // weakest-edge overflow rule all go: ~6 of the layer's 36 vector instructions per edge
#pragma unroll
                            for (int j = 0; j < DEG; j++) {
                                sx ^= __float_as_uint(x[j]);
                                u[j] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(__builtin_fmaf(fabsf(x[j]), 1.44269504088896341f, -1.f)) + 0.5f);
                            }
                            float b = 0.f;
#pragma unroll
                            for (int j = DEG - 1; j >= 0; j--) {
                                if (j % BS == BS - 1 || j == DEG - 1) B[j / BS] = b;
                                b = __builtin_fmaf(u[j], 1.f - b, b);
                            }
                        } else {
                        float mn2 = INFINITY;
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            const float a = fabsf(x[j]);
                            mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                            mn1 = fminf(mn1, a);
                            sx ^= __float_as_uint(x[j]);
                        }
                        const float s2 = fmaxf(0.f, (mn2 - 16.f) * 1.44269504088896341f);
                        kap = __builtin_amdgcn_exp2f(-s2);
                        cln = s2 * 0.693147180559945309f;
                        const float s2p1 = s2 + 1.f, hk = 0.5f * kap;
                        key = (mn2 - mn1 > 60.f) ? mn1 : __builtin_nanf("");      // a_j <> key: "this edge is not the weakest one and the weakest one overflows"
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            // u' = 2^s2 . 2 / (e^a + 1) = 1 / (2^(a log2 e - s2 - 1) + 2^-(s2 + 1)): fma, exp, add, rcp; a = +inf (absent edge) -> 0
                            const float ea = __builtin_fmaf(fabsf(x[j]), 1.44269504088896341f, -s2p1);
                            u[j] = (SPA_ABL & 8) ? ea : __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(ea) + hk);
                        }
                        {
                            float b = 0.f;      // B_j = Q' of the slots behind j; B[k] = B_{BS k + BS - 1} (clipped to the last slot)
#pragma unroll
                            for (int j = DEG - 1; j >= 0; j--) {
                                if (j % BS == BS - 1 || j == DEG - 1) B[j / BS] = b;
                                b = comb(b, u[j]);
                            }
                        }
                        }
                    }
                    if (ncf > 0) __syncthreads();         // every read of the layer precedes its writes
                    if (act) {
                        float A = 0.f;
                        const bool anykey = __ballot(key == key) != 0ull;      // (NaN stands for "no overflow in this check")
                        float mq[4] = {0.f, 0.f, 0.f, 0.f};
                        if (DEG > 13) asm volatile("" : "+v"(t4s));      // (the short codes keep their offsets: 70 registers in all)
                        if (AT16 && DEG > 13) {      // (the halves are taken apart again for the stores: 27 addresses kept across the arithmetic do not fit)
#pragma unroll
                            for (int k = 0; k < ND16; k++) asm volatile("" : "+v"(w16[k]));
                        }
                        __builtin_amdgcn_s_setprio(2);
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            float nw;                                 // magnitude bits of o under the sign of (all signs) ^ (own sign): one v_bfi_b32
                            if constexpr (SPA == 2) {
                                float val = tprod / tv[j];              // 0 / 0 = NaN takes the cap, as in the oracle
                                val = (val < 1.0f) ? val : __uint_as_float(0x3F7FFFFEu);      // 1 - FLT_EPSILON
                                const float o = w8_det_log1p((val + val) / (1.0f - val));
                                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(nw) : "s"(MAGM), "v"(o), "v"(sx ^ __float_as_uint(x[j])));
                            } else if constexpr (SPA == 3) {
                                const float wA = 1.f - A;
                                float Bj;
                                {
                                    const int js = (j / BS) * BS + BS - 1 < DEG - 1 ? (j / BS) * BS + BS - 1 : DEG - 1;
                                    Bj = B[j / BS];
#pragma unroll
                                    for (int i = js; i > j; i--) Bj = __builtin_fmaf(u[i], 1.f - Bj, Bj);
                                }
                                const float Q = __builtin_fmaf(Bj, wA, A);
                                float o = __builtin_amdgcn_logf(__builtin_fmaf(2.f, __builtin_amdgcn_rcpf(Q), -1.f)) * 0.693147180559945309f;
                                o = fminf(o, p.spa_cap);
                                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(nw) : "s"(MAGM), "v"(o), "v"(sx ^ __float_as_uint(x[j])));
                                A = __builtin_fmaf(u[j], wA, A);
                            } else {
                            const float wA = __builtin_fmaf(-kap, A, 1.f);
                            float Bj;
                            {
                                const int js = (j / BS) * BS + BS - 1 < DEG - 1 ? (j / BS) * BS + BS - 1 : DEG - 1;      // the kept slot at or behind j
                                Bj = B[j / BS];
#pragma unroll
                                for (int i = js; i > j; i--) Bj = comb(Bj, u[i]);
                            }
                            const float Q = __builtin_fmaf(Bj, wA, A);
                            // log2((2 - kap Q) / Q) as ONE logarithm of 2 / Q - kap (v_rcp + fma + v_log instead of fma + 2 v_log + subtract; 2 / Q >= 2 kap: no
                            // cancellation; same deviation from the oracle, 3.3e-6 max(1, |L|) at most; short frames +1.0-1.6 %, same-box A/B)
                            const float lg = (SPA_ABL & 8) ? Q : __builtin_amdgcn_logf(__builtin_fmaf(2.f, __builtin_amdgcn_rcpf(Q), -kap));
                            float o = __builtin_fmaf(lg, 0.693147180559945309f, cln);
#if W8_SPA_ANYKEY
                            // (round 5) the overflow rule behind a wave-uniform branch: it applies to a check whose two smallest magnitudes differ by more than 60, which no
                            // wave of a frame near the waterfall holds; as two compares, an or and a select per slot it was 4 of the layer's ~34 vector instructions per edge
                            if (anykey) { asm volatile("" ::: "memory"); o = __builtin_islessgreater(fabsf(x[j]), key) ? mn1 : o; }
#else
                            o = (fabsf(x[j]) < key || fabsf(x[j]) > key) ? mn1 : o;
#endif
                            o = fminf(o, p.spa_cap);      // (`--dec-implem SPA`: clipped where the messages of AFF3CT's tanh-product rule saturate; SPA_EXACT: +inf)
                            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(nw) : "s"(MAGM), "v"(o), "v"(sx ^ __float_as_uint(x[j])));
                            A = __builtin_fmaf(u[j], wA, A);
                            }
                            const bool pr = ((prim >> j) & 1u) != 0u;
                            if (AT16) {
                                // the table's address is where the value came from (check 0's absent p_{c-1}: the junk row's word 0); a duplicate edge's or a NULL slot's plain store
                                // (redirected to the junk row in the other form) is left out -- those are the first slots only
                                if (j >= KDS || pr) lst(a16(j), x[j] + nw);
                            } else {
                            const uint32_t base = (E[j] >> 11) & 0x3FFFFu, wj = woff_s(j);
                            if (w8_slot_lds(MODE, j)) {
                                uint32_t a = wj + (pr ? base : ljunk);
                                if (j == DEG - 1 && mask0) a = ljunk;
                                lst(a, x[j] + nw);
                            } else {
                                const uint32_t sb = (w8_hybrid(MODE) || pr) ? base : 0u;
                                const uint32_t vo = (j == DEG - 1 && mask0) ? W8_OOB : wj;
                                if (FWD && j == DEG - 2 && r + 1 < q) pfw = x[j] + nw;      // p_c: kept for layer r + 1
                                else if (SPA_ABL & 4) asm volatile("" :: "v"(x[j] + nw), "v"(vo), "s"(sb));
                                else gst(vo, sb, x[j] + nw);
                            }
                            }
#if SPA_MSG4
                            mq[j & 3] = nw;                     // four slots' messages leave as one 16-byte piece (the last group: what is left)
                            if (!(SPA_ABL & 1)) {
                                if ((j & 3) == 3)
                                    __builtin_amdgcn_raw_buffer_store_b128(m_u32x4{__float_as_uint(mq[0]), __float_as_uint(mq[1]), __float_as_uint(mq[2]), __float_as_uint(mq[3])}, rs,
                                                                           wide_off(t4s * 4u, mrow + (uint32_t)(j >> 2) * (W8_ROW * 4u)), 0u, SPA_AUX_ST);
                                else if (j == DEG - 1) {
                                    const uint32_t tbase = mrow + (uint32_t)MG4 * (W8_ROW * 4u);
                                    if (MR == 3) __builtin_amdgcn_raw_buffer_store_b96(m_u32x3{__float_as_uint(mq[0]), __float_as_uint(mq[1]), __float_as_uint(mq[2])}, rs, wide_off(t4s * 3u, tbase), 0u, SPA_AUX_ST);
                                    if (MR == 2) __builtin_amdgcn_raw_buffer_store_b64(m_u32x2{__float_as_uint(mq[0]), __float_as_uint(mq[1])}, rs, t4s * 2u, tbase, SPA_AUX_ST);
                                    if (MR == 1) mst(t4s, tbase, mq[0]);
                                }
                            } else asm volatile("" :: "v"(nw));
#else
                            mst(t4s, mrow + (uint32_t)j * mpitch, nw);
#endif
                            if (OD_STATIC) { if (j < LDPC_SPA_MAXC) od[j] = nw - od[j]; }
                            else if ((dupmask >> j) & 1u) {          // wave-uniform
#pragma unroll
                                for (int i = 0; i < LDPC_SPA_MAXC; i++) if (i < ncf && dup_slot(i) == (uint32_t)j) od[i] = nw - od[i];
                            }
                        }
                        if (AT16) {      // the next layer's addresses: this layer's stores have read theirs; they travel under the replay and the end barrier
                            __builtin_amdgcn_sched_barrier(0);
                            at16_request(r + 1 < q ? r + 1 : 0);
                        }
#if SPA_PREFETCH
                        // the messages of the next layer do not depend on this one: their loads (HBM misses, the message store is
                        // streamed through once per iteration) go out now and travel while this layer's stores drain at the barriers
                        __builtin_amdgcn_sched_barrier(0);      // not earlier: x[], u[], B[] have to be dead first (registers)
                        if ((it > 0 || r + 1 == q) && !(r + 1 == q && it + 1 >= p.n_ite)) {
                            const uint32_t mnext = st_base + (uint32_t)((r + 1 < q ? r + 1 : 0) * DEG) * mpitch;
#if SPA_MSG4
                            mgrp_ld(onx, t4s, mnext);
#else
#pragma unroll
                            for (int j = 0; j < DEG; j++) onx[j] = mld(t4s, mnext + (uint32_t)j * mpitch);
#endif
                        } else {
#pragma unroll
                            for (int j = 0; j < DEG; j++) onx[j] = 0.f;
                        }
#endif
                        __builtin_amdgcn_s_setprio(0);
                    }
                    // duplicate edges: level by level behind a barrier, as in the min-sum layer
                    uint32_t prev_lvl = 0u;
#pragma unroll
                    for (int i = 0; i < LDPC_SPA_MAXC; i++) {
                        if (i >= ncf) break;
                        const uint32_t e = i == 0 ? ce0 : i == 1 ? ce1 : T[32 + i];
                        const uint32_t lvl = i == 0 ? 1u : i == 1 ? (cinfo >> 21) & 3u : T[48 + i] >> 8;
                        if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
                        if (act) {
                            const uint32_t d = t4 - (e & 0x7FFu);
                            const uint32_t off = min(d, d + (uint32_t)W8_ROW), base = (e >> 11) & 0x3FFFFu;
                            if (MODE != 1) { const float Lv = lld(off + base); lst(off + base, Lv + od[i]); }
                            else { const float Lv = gld(off, base); gst(off, base, Lv + od[i]); }
                        }
                    }
                    if (TEV) tev = p.w8.tab[(r + 1 < q ? r + 1 : 0) * LDPC_FAST_STRIDE + (w8_lane_now() & 31)];
                    else {
                        const const_u32 Tn = tab + (r + 1 < q ? r + 1 : 0) * LDPC_FAST_STRIDE;
#pragma unroll
                        for (int j = 0; j < 32; j++) TE[j] = Tn[j];
                    }
                    __syncthreads();
                    continue;
                }
                float v[DEG];
                // (LDS-only image only -- same-box A/B: short frames 4.51 -> 4.44 ms per 16384, the hybrid image of the normal frames 5.77 -> 5.85 per 4096 with them)
                // (LDS-only image only.  On the hybrid images -- global slots' loads first, LDS slots' behind them, barrier once the LDS data are back -- it is 2.5-5.7 % SLOWER, same-box
                // A/B 5.94-6.04 against 5.62-5.70 ms: a barrier at the layer's start keeps the waves in phase, and in-phase waves want the issue port at the same moment)
                constexpr bool EARLY_B1 = W8_EARLY_B1 && MODE == 0;
                // (the early barrier is an s_barrier inside the block of the active lanes, matched by a hand-written one for the waves without checks: every WORKING wave has to have
                //  an active lane -- a wave whose lanes are all idle would skip the block, and the barrier with it)
                static_assert(LDPC_Z > 64 * 5 && LDPC_Z <= 64 * 6, "six working waves, the last one with active lanes (W8_EARLY_B1's barrier sits inside `if (act)`)");
                constexpr bool ABSF = W8_ABS_FOLD && MODE == 0, SGNA = W8_SIGN_ADD && MODE == 0;
                constexpr int KD = MODE == 0 ? ldpc_w8_kd(DEG) : DEG;       // LDS-only image: duplicate edges sit in slots < KD (plan), the others are primary
                // (round 4) LDS-only image: conflict entry i is slot i < KDD (plan), so what a duplicate edge adds in the replay, new - old message, is kept from the passes
                // (one subtraction per slot) instead of being rebuilt from the packed states behind the barrier (two unpacks = 8 vector instructions per entry on the
                // critical path between the layer's last two barriers: the replay was 18 % of a short-frame launch).  The hybrid images have no registers for it.
                constexpr bool DREG = W8_DELTA_REG && MODE == 0;
                constexpr int KDD = ldpc_w8_kd(DEG);
                float dold[KDD];
#pragma unroll
                for (int j = 0; j < KDD; j++) dold[j] = 0.f;
                const float c1o = nx1, c2o = nx2;
                const uint32_t pko = __float_as_uint(nxk);
                float mn1 = INFINITY, mn2 = INFINITY, cst1 = 0.f, cst2 = 0.f;
                uint32_t sacc = 0u, tot = 0u, pkn = 0u;
                if (EARLY_B1 && ncf > 0 && role < 0) asm volatile("s_barrier" ::: "memory");      // (the waves without checks of the modes without parked rows)
                if (act) {
                    // ---- pass 1a: every posterior load of the check in flight before any use.  Wave priorities (same-box A/B, tools/ab_kernel.sh):
                    // 3 while the loads are issued, 0 while pass 1b waits for them, 2 in pass 2 (stores, and the way to the barrier) is
                    // worth 2.5 %; (3, 0, 3) 1.8 %, (3, 1, 2) 1.4 %, load issue alone 1.2 %, the replay stretch on top nothing.  Round 3, LDS-only image (short
                    // frames): 3 in pass 2 and 1 behind it (replay, end barrier) instead of 2 and 0: QPSK-S 8/9 5.68 -> 5.47 ms, 3/5 8.86 -> 8.75 ms per 16384 frames;
                    // the hybrid image keeps (3, 0, 2, 0): with (3, 0, 2 | 3, 1) its steady state gains 1.4-2.1 % and the 4096-frame launch of the bench loses 0.3-0.8 %
                    __builtin_amdgcn_s_setprio(3);
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        const uint32_t base = (E[j] >> 11) & 0x3FFFFu;
                        if (ATAB && (MODE == 0 || w8_slot_lds(MODE, j))) {
                            // (round 4) the address comes from the per-lane table (requested behind the previous layer's last store): an LDS slot's entry is the whole
                            // LDS address, a global slot's the rotated offset inside its row -- no vector instruction per slot (they were 21 % of the layer's vector issue
                            // cycles: three instructions per slot, two of them at the 4.25-cycle price of an SGPR operand, four for an LDS slot, every iteration again)
                            if (FWD && j == DEG - 1) { if (r > 0) v[j] = pfw; else v[j] = gld1(w[j], base); }
                            else v[j] = w8_slot_lds(MODE, j) ? lld(w[j]) : gld1(w[j], base);
                            continue;
                        }
                        const uint32_t d = t4 - (E[j] & 0x7FFu);
                        w[j] = min(d, d + (uint32_t)W8_ROW);
                        if (FWD && j == DEG - 1) { if (r > 0) v[j] = pfw; else v[j] = gld1(w[j], base); }      // p_{c-1}: handed over by layer r - 1
                        else if (w8_slot_lds(MODE, j)) {
                            const uint32_t a = w[j] + base;
                            v[j] = lld(a);
                            if (j >= KD) w[j] = a;            // a primary edge by the plan's slot order: pass 2 stores where this came from
                        } else v[j] = gld1(w[j], base);
                    }
                    const int rn = r + 1 < q ? r + 1 : 0;
                    if (it == 0 && r + 1 < q) { nx1 = 0.f; nx2 = 0.f; nxk = 0.f; }       // layer r + 1 has no messages yet in the first iteration
                    else {      // {c1, c2, pk} of a check are 12 consecutive bytes: one load
                        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                        const u32x3 sv = __builtin_amdgcn_raw_buffer_load_b96(rs, t4 * 3u, st_base + (uint32_t)(rn * LDPC_Z) * 12u, 0);
                        nx1 = __uint_as_float(sv.x); nx2 = __uint_as_float(sv.y); nxk = __uint_as_float(sv.z);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    // every read of a row with duplicate edges precedes the layer's writes: those rows are LDS rows, so "the LDS loads of every wave are back" is enough; the global
                    // rows' loads stay in flight across the barrier (no vmcnt wait: __syncthreads would drain them).  Inside the block of the active lanes (a working wave always has
                    // some): closing the block here would split pass 1a / 1b into two scheduling regions, which alone costs 3-4 % (same-box A/B)
                    if (EARLY_B1 && ncf > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    PROF_MARK(0);
                    // ---- pass 1b: v->c = posterior - old c->v ; running min1 / min2 / signs
                    const uint32_t idxo = pko >> 27;
                    if (it == 0) {
                        // first iteration: there are no messages yet, v->c is the posterior itself (4 VALU per edge less)
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            float x = v[j];
                            if (j == DEG - 1 && mask0) x = INFINITY;
                            v[j] = x;
                            const float a = fabsf(x);
                            mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                            if (ABSF) asm("v_min_f32_e64 %0, %0, |%1|" : "+v"(mn1) : "v"(x)); else mn1 = fminf(mn1, a);
                            sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(x), 31);
                        }
                    } else {
                    uint32_t psh = pko << (32u - DEG);      // (W8_SIGN_ADD) sign bit of slot j in bit 31 at slot j
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        const float mag = (idxo == (uint32_t)j) ? c1o : c2o;
                        const float old = and_or(SGNA ? psh : pko << ((32u - DEG) + j), SB, mag);          // sign bit of slot j | magnitude (>= 0)
                        if (SGNA && j + 1 < DEG) asm("v_add_u32_e32 %0, %0, %0" : "+v"(psh));      // (written out: the compiler turns p + p back into a shift)
                        if (DREG && j < KDD) dold[j] = old;
                        float x = v[j] - old;
                        if (j == DEG - 1 && mask0) x = INFINITY;
                        v[j] = x;
                        const float a = fabsf(x);
                        if (W8_ABL & 4) { if (j == 0) { mn1 = a; mn2 = a + 1.f; sacc = __float_as_uint(x); } continue; }
                        mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                        if (ABSF) asm("v_min_f32_e64 %0, %0, |%1|" : "+v"(mn1) : "v"(x)); else mn1 = fminf(mn1, a);
                        sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(x), 31);      // shift the sign bit in
                    }
                    }
                    cst1 = mn2 * p.alpha; cst2 = mn1 * p.alpha;
                    tot = (uint32_t)(__popc(sacc) & 1);                                       // parity of all signs
                    pkn = sacc ^ (tot ? ((1u << DEG) - 1u) : 0u);                             // sign(new_j) = tot ^ sign(x_j)
                }
                PROF_MARK(1);
                if (!EARLY_B1 && ncf > 0) __syncthreads();         // every read of the layer precedes its writes
                PROF_MARK(2);
                if (act) {
                    __builtin_amdgcn_s_setprio(MODE == 0 ? 3 : 2);
                    // ---- pass 2: new c->v ; posterior = v->c + new c->v.  Duplicate edges (not in `prim`) go to
                    //      the junk row (selected on the scalar unit), the absent edge of lane 0 is dropped.
                    uint32_t idxn = 0u;
                    float m1s = __uint_as_float(__float_as_uint(cst1) | (tot << 31));        // output magnitudes carrying the total sign
                    float m2s = __uint_as_float(__float_as_uint(cst2) | (tot << 31));
                    asm volatile("" : "+v"(m1s), "+v"(m2s));      // keep the sign folded in: one select + one bit-op per edge
                    constexpr int P2_ROT = (W8_P2_GFIRST && w8_hybrid(MODE) && !ATAB) ? (MODE == 3 ? W8_NL : ldpc_park_nl(MODE)) : 0;      // first slot of pass 2 (the first global one)
#pragma unroll
                    for (int jj = 0; jj < DEG; jj++) {
                        const int j = jj + P2_ROT < DEG ? jj + P2_ROT : jj + P2_ROT - DEG;
                        const float x = v[j];
#if W8_IDX_E32
                        // A SIMD issues a VOP3-encoded instruction every ~4.1 cycles and a VOP2 / VOPC one every ~2.1, whatever the number of waves (tools/probe_issue2.hip):
                        // the two selects on the comparison are written so that both take the 32-bit encoding (D = vcc ? src1 : src0 with src1 a register: the test is
                        // "not the minimum", the slot number the inline constant in src0); the compiler's form of the second one is v_cndmask_b32_e64.
                        float mag;
                        asm("v_cmp_neq_f32_e64 vcc, |%2|, %3\n\tv_cndmask_b32_e32 %0, %4, %5, vcc\n\tv_cndmask_b32_e32 %1, %6, %1, vcc"
                            : "=&v"(mag), "+v"(idxn) : "v"(x), "v"(mn1), "v"(m1s), "v"(m2s), "n"(j) : "vcc");
                        const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & SB));
#else
                        const bool ismin = (W8_ABL & 8) ? j == 3 : fabsf(x) == mn1;
                        const float mag = ismin ? m1s : m2s;
                        const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & SB));
                        idxn = ismin ? (uint32_t)j : idxn;
                        asm("" : "+v"(idxn));                     // select now: the comparison mask dies here instead of piling up 27 SGPR pairs
#endif
                        if (DREG && j < KDD) dold[j] = nw - dold[j];
                        const bool pr = ((prim >> j) & 1u) != 0u;                             // wave-uniform
                        const uint32_t base = (E[j] >> 11) & 0x3FFFFu;
                        if (w8_slot_lds(MODE, j) && ATAB) {      // (hybrid images: every LDS slot is covered by the table's pieces)
                            // the table's address is where the value came from; a duplicate edge's plain store (redirected to the junk row without the table) is left out
                            uint32_t a = w[j];
                            if (j == DEG - 1 && mask0) a = ljunk;
                            if ((MODE == 0 && j >= KD) || pr) lst(a, x + nw);
                            if (MODE != 0 && j == AT_NS - 1 && role >= 0) {
                                // hybrid image: the next layer's LDS addresses are requested HERE, behind the last LDS slot's store and in front of the global slots' stores --
                                // their registers are free, and the wait for them does not have to sit out the stores' acknowledgements (vmcnt counts in order)
                                __builtin_amdgcn_sched_barrier(0);
                                at_request(r + 1 < q ? r + 1 : 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else if (w8_slot_lds(MODE, j)) {
                            uint32_t a = j >= KD ? w[j] : w[j] + (pr ? base : ljunk);
                            if (j == DEG - 1 && mask0) a = ljunk;
                            lst(a, x + nw);
                        } else {
                            // MODE 3: the duplicate edges all live in LDS, a global slot is always primary
                            const uint32_t sb = (w8_hybrid(MODE) || pr) ? base : 0u;        // global junk row = row 0
                            const uint32_t vo = (j == DEG - 1 && mask0) ? W8_OOB : w[j];
                            if (FWD && j == DEG - 2) { if (r + 1 < q) pfw = x + nw; else gst2(vo, sb, x + nw); }      // p_c: kept for layer r + 1
                            else gst2(vo, sb, x + nw);
                        }
                    }
                    pkn |= idxn << 27;
                    {
                        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                        u32x3 sv; sv.x = __float_as_uint(cst1); sv.y = __float_as_uint(cst2); sv.z = pkn;
                        __builtin_amdgcn_raw_buffer_store_b96(sv, rs, wide_off(t4 * 3u, st_base + (uint32_t)(r * LDPC_Z) * 12u), 0u, NMS_AUX_ST);
                    }
                    if (q == 1) { nx1 = cst1; nx2 = cst2; nxk = __uint_as_float(pkn); }
                    __builtin_amdgcn_s_setprio(MODE == 0 ? 1 : 0);
                }
                if (ATAB && MODE == 0 && role >= 0) {      // the next layer's addresses: requested now (this layer's stores have read theirs), they travel under the replay and the end barrier
                    __builtin_amdgcn_sched_barrier(0);
                    at_request(r + 1 < q ? r + 1 : 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                PROF_MARK(3);
                // ---- duplicate edges of a bit-group inside this layer: ordered delta updates, level by level.  The
                //      first two (nearly always all of them, both of level 1) travel with the layer table.
                if (ncf > 0) {
                    auto addr_of = [&](uint32_t e) __attribute__((always_inline)) { const uint32_t d = t4 - (e & 0x7FFu); return min(d, d + (uint32_t)W8_ROW); };
                    auto delta_of = [&](uint32_t j) __attribute__((always_inline)) { return w8_unpack<DEG>(cst1, cst2, pkn, j, SB) - w8_unpack<DEG>(c1o, c2o, pko, j, SB); };
                    const uint32_t j0 = (cinfo >> 8) & 31u, j1 = (cinfo >> 16) & 31u, lvl1 = (cinfo >> 21) & 3u;
                    const bool two = ncf > 1 && lvl1 == 1u;         // entry 1 commutes with entry 0 (another bit-group)
                    // (round 4, LDS-only image) addresses and deltas of the first two entries are ready BEFORE the barrier: behind it only load, add, store remain
                    // (the hybrid images keep round 3's order -- everything behind the barrier: hoisting it measured 1 % SLOWER there, 5.72 against 5.67 ms)
                    uint32_t o0 = 0, b0 = 0, o1 = 0, b1 = 0;
                    float d0 = 0.f, d1 = 0.f;
                    if (DREG) { o0 = addr_of(ce0); b0 = (ce0 >> 11) & 0x3FFFFu; o1 = addr_of(ce1); b1 = (ce1 >> 11) & 0x3FFFFu; d0 = dold[0]; d1 = dold[KDD > 1 ? 1 : 0]; }
                    __syncthreads();                                // the primary writes of the layer are in place
                    if (act) {
                        if (!DREG) { o0 = addr_of(ce0); b0 = (ce0 >> 11) & 0x3FFFFu; o1 = addr_of(ce1); b1 = (ce1 >> 11) & 0x3FFFFu; d0 = delta_of(j0); d1 = delta_of(j1); }
                        float L0, L1 = 0.f;
                        if (MODE != 1) { L0 = lld(o0 + b0); if (two) L1 = lld(o1 + b1); }
                        else { L0 = gld(o0, b0); if (two) L1 = gld(o1, b1); }
                        const float n0 = L0 + d0, n1 = L1 + d1;
                        if (MODE != 1) { lst(o0 + b0, n0); if (two) lst(o1 + b1, n1); }
                        else { gst(o0, b0, n0); if (two) gst(o1, b1, n1); }
                    }
                    uint32_t prev_lvl = 1u;
                    if (DREG) {
#pragma unroll
                        for (int i = 1; i < KDD; i++) {
                            if (i < (two ? 2 : 1) || i >= ncf) continue;      // (no `break`: the loop has to unroll completely, dold[] lives in registers)
                            const uint32_t e = T[32 + i], lvl = T[48 + i] >> 8;
                            const uint32_t off = addr_of(e), base = (e >> 11) & 0x3FFFFu;
                            if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
                            if (act) { const float Lv = lld(off + base); lst(off + base, Lv + dold[i]); }
                        }
                    } else
                    for (int i = two ? 2 : 1; i < ncf; i++) {
                        const uint32_t e = T[32 + i], meta = T[48 + i];
                        const uint32_t j = meta & 31u, lvl = meta >> 8;
                        if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
                        if (act) {
                            const uint32_t off = addr_of(e), base = (e >> 11) & 0x3FFFFu;
                            if (MODE != 1) { const float Lv = lld(off + base); lst(off + base, Lv + delta_of(j)); }
                            else { const float Lv = gld(off, base); gst(off, base, Lv + delta_of(j)); }
                        }
                    }
                }
                PROF_MARK(4);
                {
                    const const_u32 Tn = tab + (r + 1 < q ? r + 1 : 0) * LDPC_FAST_STRIDE;
#pragma unroll
                    for (int j = 0; j < 32; j++) TE[j] = Tn[j];
                }
                __syncthreads();
                PROF_MARK(5);
            }
            it++;
            if (p.early_stop || it == p.n_ite) {
                // ---- syndrome of the hard decisions (this lane's check of every layer), layer by layer: the posteriors are final, so the
                //      order is free and the first layer with an unsatisfied check settles the answer -- a frame that has not converged
                //      (every iteration but the last of a frame under the reference's stopping rule) pays for one layer instead of q.
                //      Same-box A/B: the simulator with the early stop +13 % (same frame errors).  With a fixed number of iterations there is
                //      one vote, after the last layer, as before (a vote per layer in that mode too made the bench 0.4 % slower)
                ok = true;
                int bad = 0;
                // the workgroup's vote as a ballot per wave + one LDS word out of three in rotation (zeroed two votes ahead), one barrier per vote
                // (__syncthreads_or brings 256 bytes of static LDS: the dynamic block then starts at 0x100 and every LDS address the layer loop keeps in
                // a register needs that offset added on the vector unit; it also keeps the 64-bit thread index alive across the layer loop)
                auto vote = [&](int b) __attribute__((always_inline)) -> bool {
                    const bool any = __ballot(b != 0) != 0ull;
                    lds_int *const w = s_misc + 10;
                    const int k = nvote % 3;
                    nvote++;
                    if (w8_lane_now() == 0) { if (any) w[k] = 1; if (wave == 0) w[(k + 1) % 3] = 0; }
                    __syncthreads();
                    return w[k] != 0;
                };
                for (int r = 0; r < q; r++) {
                    if (act) {
                        const const_u32 T = tab + r * LDPC_FAST_STRIDE;
                        float Lv[DEG];
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            const uint32_t e = T[j];
                            const uint32_t d = t4 - (e & 0x7FFu);
                            const uint32_t wo = min(d, d + (uint32_t)W8_ROW), base = (e >> 11) & 0x3FFFFu;
                            Lv[j] = w8_slot_lds(MODE, j) ? lld(wo + base) : gld(wo, base);
                        }
                        if (r == 0 && t == 0) Lv[DEG - 1] = 0.f;                               // absent edge
                        uint32_t x = 0u;
#pragma unroll
                        for (int j = 0; j < DEG; j++) x ^= __float_as_uint(Lv[j]);             // NULL slots read +inf
                        bad |= (int)(x >> 31);
                    }
                    if (w8_parked(MODE)) {
                        // parked rows: the idle waves move rows for layer r + 1 while this layer is read, so every layer ends at a barrier; under the
                        // stopping rule layer 0 is voted on BEFORE anything moves (a frame that has not converged nearly always fails there, and
                        // the schedule then stands where the next iteration needs it), otherwise the idle waves run the schedule to its end
                        if (p.early_stop) {
                            if (vote(bad)) { ok = false; if (r > 0) __syncthreads(); break; }
                            if (r == 0) __syncthreads();
                        } else if (r == q - 1) { if (vote(bad)) ok = false; }
                        else __syncthreads();
                        continue;
                    }
                    if ((p.early_stop || r == q - 1) && vote(bad)) { ok = false; break; }
                }
                PROF_MARK(6);
                if (ok) break;
            }
        }

        // ---- outputs: hard decisions of the info bits (image rows in storage order, W8_IO loads in flight)
        const int lo = SPA ? w8_lane_now() : lane, to = SPA ? role * 64 + lo : t;
        const bool first = SPA ? wave == 0 && lo == 0 : threadIdx.x == 0;
        if (first) {
            if (p.cwd) p.cwd[f] = ok ? 1 : 0;
            if (p.ites) p.ites[f] = it;
        }
        const int n_words = (p.K + 31) / 32;
        const const_u32 prbs_c = (const_u32)p.info_prbs;
        auto emit = [&](int g, float Lv) __attribute__((always_inline)) {
            if (p.info_out && role >= 0 && g < p.n_info) {
                // fused chain: descrambled info bits as int32, the BCH stage's output for a frame it leaves alone.  The 64 PRBS bits
                // of this wave's stretch of the row (bit 360 g + 64 role onwards) come in as ONE wave-uniform 64-bit scalar word of a
                // table laid out by (row, wave).
                const int kb = g * LDPC_Z + role * 64;
                const const_u32 pq = prbs_c + 2 * (g * 6 + role);
                const unsigned long long m64 = (unsigned long long)pq[0] | ((unsigned long long)pq[1] << 32);
                const int k = kb + lo;
                if (act && k < p.K_info)
                    __builtin_nontemporal_store((int32_t)((Lv < 0.f ? 1u : 0u) ^ (uint32_t)((m64 >> lo) & 1ull)), &p.info_out[(size_t)f * p.K_info + k]);
            }
            if (act) {
                if (g < p.n_info) {
                    if (p.bits) __builtin_nontemporal_store((int32_t)(Lv < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + g * LDPC_Z + to]);
                    if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + to] = Lv;
                } else if (p.post) p.post[(size_t)f * p.N + p.K + q * to + (g - p.n_info)] = Lv;
            }
            if (p.packed && role >= 0 && g < p.n_info) {
                // 64 lanes = 64 consecutive info bits starting at bit o = 360 g + 64 role of the frame: o is a multiple of 8, so the
                // ballot is 8 whole bytes (5 for the last wave of a row) that no other wave or row touches -- plain byte stores,
                // no atomics and no zeroing of the image
                const unsigned long long m = __ballot(act && Lv < 0.f);
                const int o8 = g * (LDPC_Z / 8) + role * 8, nb = role * 64 + 64 <= LDPC_Z ? 8 : (LDPC_Z - role * 64) / 8;
                if (lo < nb) reinterpret_cast<uint8_t *>(p.packed + (size_t)f * n_words)[o8 + lo] = (uint8_t)(m >> (8 * lo));
            }
        };
        const int nl_out = p.post ? nl : nl_info, ng_out = p.post ? ng : ng_info;
        // (round 4) The two output forms every production call uses get loops of their own: `emit` tests four wave-uniform pointers per row and forms 64-bit
        // addresses per lane (the frame index of the min-sum kernel lives in a vector register), ~110 instructions per row and wave -- the output phase was 7 % of
        // the launch, as long as 17 layers.  Here the frame's output block is a buffer descriptor made once per frame on the scalar unit (the row offset goes into the
        // instruction's scalar offset, a lane beyond the block is dropped by the range check: no compare, no address arithmetic on the vector unit) and the chain's
        // descrambling is ONE scalar XOR of the wave's ballot with the 64 PRBS bits of its stretch of the row.
        const int fu = __builtin_amdgcn_readfirstlane(f);
        const bool out_plain = p.bits && !p.post && !p.info_out && !p.packed, out_chain = p.info_out && p.packed && !p.bits && !p.post;
        const uint32_t vo_out = act ? (uint32_t)to * 4u : W8_OOB;      // (the lanes past the 360th check store nowhere)
        const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(out_plain ? (void *)(p.bits + (size_t)fu * p.K) : (void *)gwork, 0, out_plain ? p.K * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_info = __builtin_amdgcn_make_buffer_rsrc(out_chain ? (void *)(p.info_out + (size_t)fu * p.K_info) : (void *)gwork, 0, out_chain ? p.K_info * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_pack = __builtin_amdgcn_make_buffer_rsrc(out_chain ? (void *)(p.packed + (size_t)fu * n_words) : (void *)gwork, 0, out_chain ? n_words * 4 : 0, 0x00020000);
        const int nb_w = role * 64 + 64 <= LDPC_Z ? 8 : (LDPC_Z - role * 64) / 8;      // whole bytes of a row this wave's ballot holds
        const uint32_t pk_vo = (role >= 0 && lo < nb_w) ? (uint32_t)(role * 8 + lo) : W8_OOB, pk_sh = (uint32_t)(lo & 3) * 8u;
        const bool pk_hi = lo >= 4;
        auto emit_plain = [&](int g, float Lv) __attribute__((always_inline)) {       // bits socket alone (dvbs2hip_ldpc_decode_siho*, the bench line)
            __builtin_amdgcn_raw_buffer_store_b32(Lv < 0.f ? 1u : 0u, rs_bits, vo_out, (uint32_t)g * (uint32_t)W8_ROW, 2);
        };
        auto emit_chain = [&](int g, float Lv) __attribute__((always_inline)) {       // fused chain: packed hard decisions for the BCH stage + descrambled information bits as int32 (rows of information groups only)
            const unsigned long long m = __ballot(Lv < 0.f);                      // (inactive lanes hold 0.f)
            const const_u32 pq = prbs_c + 2 * (g * 6 + role);
            const unsigned long long d = m ^ ((unsigned long long)pq[0] | ((unsigned long long)pq[1] << 32));
            uint32_t bit;
            asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(bit) : "s"(d));             // this lane's bit of the descrambled word (the 64-bit mask is the select's condition)
            // (the whole offset in the vector register: the range check that drops the bits behind K_info then sees all of it)
            if (!(W8_ABL & 32)) __builtin_amdgcn_raw_buffer_store_b32(bit, rs_info, vo_out + (uint32_t)g * (uint32_t)W8_ROW, 0u, 2);
            const uint32_t half = pk_hi ? (uint32_t)(m >> 32) : (uint32_t)m;
            if (!(W8_ABL & 16)) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(half >> pk_sh), rs_pack, pk_vo, (uint32_t)g * (uint32_t)(LDPC_Z / 8), 0);
        };
        auto run_out = [&](auto &&em) __attribute__((always_inline)) {
            if (role >= 0) {
                for (int l0 = 0; l0 < nl_out; l0 += W8_IO) {
                    float v[W8_IO];
#pragma unroll
                    for (int k = 0; k < W8_IO; k++) v[k] = act ? lld((uint32_t)(l0 + k < nl_out ? l0 + k : nl_out - 1) * W8_ROW + t4) : 0.f;
#pragma unroll
                    for (int k = 0; k < W8_IO; k++) if (l0 + k < nl_out) em((int)rows[l0 + k], v[k]);
                }
                for (int l0 = 0; l0 < ng_out; l0 += W8_IO) {
                    float v[W8_IO];
#pragma unroll
                    for (int k = 0; k < W8_IO; k++) v[k] = act ? gld(t4, grow0 + (uint32_t)(l0 + k < ng_out ? l0 + k : ng_out - 1) * W8_ROW) : 0.f;
#pragma unroll
                    for (int k = 0; k < W8_IO; k++) if (l0 + k < ng_out) em((int)rows[nl + l0 + k], v[k]);
                }
            }
            if (w8_parked(MODE)) {
                constexpr int NRP = ldpc_park_nr(MODE);
                // the rows parked in the idle waves' registers: handed over through LDS positions 0 .. NR-1 (every LDS row has been read)
                __syncthreads();
                __syncthreads();
                const const_u32 srow = rows + nl + ng + q;
                for (int l0 = 0; l0 < NRP; l0 += W8_IO) {
                    float v[W8_IO];
#pragma unroll
                    for (int k = 0; k < W8_IO; k++) v[k] = act ? lld((uint32_t)(l0 + k < NRP ? l0 + k : NRP - 1) * W8_ROW + t4) : 0.f;
#pragma unroll
                    for (int k = 0; k < W8_IO; k++) if (l0 + k < NRP && srow[l0 + k] != 0xFFFFFFFFu) em((int)srow[l0 + k], v[k]);
                }
            }
        };
        // (round 5) Fused chain with the BCH verification in this kernel (`syn_tab`): a frame is a BCH codeword iff r(x) mod g(x) = 0, and the remainder is linear in the
        // bits.  Bit t of row g is the coefficient of x^(360 (G - 1 - g)) x^(359 - t) (G rows of 360 information bits): the first factor A_g = x^(360 (G - 1 - g)) mod g(x)
        // depends on the row alone -- 4 or 6 words that arrive as wave-uniform scalars, like the PRBS bits -- so a lane XORs A_g into its accumulator for every row whose
        // hard decision it holds set (one mask + one and-xor per word: no vector-memory access, nothing in the way of the row loads), and the second factor is applied ONCE
        // per frame: the wave folds its lanes Horner-fashion (block of s lanes times x^s, plus the next block: six shuffle steps) into V_w = sum_l acc_l x^(63 - l),
        // reduces V_w x^(64 (5 - w)) with a small table of x^k mod g(x) and XORs the result into the workgroup's words.  What is tested is the remainder times x^24 (the
        // last wave holds 40 checks); g(0) = 1, so it is zero iff the remainder is.  The BCH stage then runs over the flagged frames only (> 99 % of the frames behind a
        // converged LDPC decoder are codewords) and rebuilds their bit image from the information bits written here plus the last row's packed bytes (the BCH parity
        // bits): the packed bytes of the other rows are not written any more.
        const bool out_syn = out_chain && p.syn_tab && p.bch_flag;
        const bool syn6 = p.syn_words > 4;
        // Per row in STORAGE order (LDS rows, global rows, register slots: the order the rows are emitted in) the plan supplies one 32-byte record {byte offset of the row in
        // the int32 socket (beyond every socket: an empty register slot), last-row flag, A_g[0 .. 5]} and the PRBS words by (storage row, wave): every scalar of a row comes
        // from an address known before the batch starts, nothing waits for the row number first (round 4's form: row number, then the PRBS word at an address formed from it).
        typedef uint32_t o_u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t o_u32x2 __attribute__((ext_vector_type(2)));
        uint32_t sacc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        const unsigned long long actm = __ballot(act);                              // the lanes that hold a check
        // (round 5, second form) The rows' scalars do not come through the scalar memory path at all: for a batch of B <= 16 rows lane l (mod 16) loads row l's 32-byte record
        // and every lane ONE dword of PRBS bits (bit k = the descrambler's bit of this lane's information bit in row k of the batch) -- three vector loads per batch, issued with
        // the batch's row loads -- and a row's scalars are read out of those registers with v_readlane.  With s_load per row (four rows' worth requested at a time: the scalar
        // file has no room for more) every group of four rows waited for a scalar round trip, the chain's output rows cost 325 cycles against the bits socket's 100.
        const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc(out_syn ? (void *)p.syn_tab : (void *)gwork, 0, out_syn ? p.syn_rows * 32 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_pl = __builtin_amdgcn_make_buffer_rsrc(out_syn ? (void *)p.info_prbs_s : (void *)gwork, 0, out_syn ? p.syn_rows * LDPC_AT_LANES * 4 : 0, 0x00020000);
        const uint32_t rec_vo = (uint32_t)(lo & 15) * 32u, pl_vo = act ? (uint32_t)to * 4u : W8_OOB;
        auto emit_syn = [&](int kk, float Lv, bool repeat, const o_u32x4 r0, const o_u32x4 r1, const uint32_t pw, auto nw_c) __attribute__((always_inline)) {      // kk: row of the batch; repeat (wave-uniform): a row that was emitted already
            constexpr int NW = decltype(nw_c)::value;
            const uint32_t off = repeat ? W8_OOB : (uint32_t)__builtin_amdgcn_readlane((int)r0.x, kk);
            const unsigned long long m = __ballot(Lv < 0.f) & (off == W8_OOB ? 0ull : actm);     // (the lanes without a check hold anything)
            uint32_t hd, mk;
            asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(hd) : "s"(m));
            asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(mk) : "s"(m));            // all ones where the hard decision is 1 (Lv < 0, as the stored bit: not the sign bit -- a posterior of -0.f decides 0)
            const uint32_t bit = hd ^ ((pw >> kk) & 1u);
            // (the whole offset in the vector register: the range check that drops the bits behind K_info, the lanes without a check and a skipped row then sees all of it)
            __builtin_amdgcn_raw_buffer_store_b32(bit, rs_info, vo_out + off, 0u, 2);
            const uint32_t A0 = (uint32_t)__builtin_amdgcn_readlane((int)r0.z, kk), A1 = (uint32_t)__builtin_amdgcn_readlane((int)r0.w, kk),
                           A2 = (uint32_t)__builtin_amdgcn_readlane((int)r1.x, kk), A3 = (uint32_t)__builtin_amdgcn_readlane((int)r1.y, kk);
            // acc ^= mask & A_g[i] as ONE instruction that takes the scalar where it is (written as `acc ^= mk & a` the compiler forms the products of a whole batch first and spills them)
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(sacc[0]) : "v"(mk), "s"(A0));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(sacc[1]) : "v"(mk), "s"(A1));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(sacc[2]) : "v"(mk), "s"(A2));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(sacc[3]) : "v"(mk), "s"(A3));
            if (NW > 4) {
                const uint32_t A4 = (uint32_t)__builtin_amdgcn_readlane((int)r1.z, kk), A5 = (uint32_t)__builtin_amdgcn_readlane((int)r1.w, kk);
                asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(sacc[NW > 4 ? 4 : 0]) : "v"(mk), "s"(A4));
                asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(sacc[NW > 4 ? 5 : 0]) : "v"(mk), "s"(A5));
            }
            if (__builtin_amdgcn_readlane((int)r0.y, kk) && !repeat) {               // the row that holds the BCH parity bits: its packed bytes for the BCH stage (row g at byte 45 g = offset / 32)
                const uint32_t half = pk_hi ? (uint32_t)(m >> 32) : (uint32_t)m;
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(half >> pk_sh), rs_pack, pk_vo, off >> 5, 0);
            }
        };
        auto fold_syn = [&](auto nw_c) __attribute__((always_inline)) {
            constexpr int NW = decltype(nw_c)::value, NU = NW + 2;
            const int los = w8_lane_now();
            // the value of lane (l ^ sh): ds_bpermute by the opaque lane index (__shfl_xor goes through __lane_id(), which the compiler computes once at the kernel's start and keeps)
            auto shx = [&](uint32_t x, int sh) __attribute__((always_inline)) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((los ^ sh) << 2, (int)x); };      // (re-formed here: every per-lane constant of this once-per-frame stretch would otherwise be hoisted out of the frame loop and live across the layer loop)
            // V_w = sum over the wave's lanes of acc_l x^(63 - l): a block of s lanes holds the sum of its lanes with the block's last lane at x^0; two neighbouring
            // blocks merge as (lower block) x^s + (upper block).  NU words hold the 32 NW + 63 bits.
            // (the reduction table's entries are requested first: their addresses do not depend on V_w, they travel under the shuffles)
            const __amdgpu_buffer_rsrc_t rs_red = __builtin_amdgcn_make_buffer_rsrc((void *)(p.syn_tab + p.syn_rows * 8), 0, LDPC_SYN_RED * NW * 4, 0x00020000);
            o_u32x4 ta[NU / 2]; o_u32x2 tb[NU / 2];
#pragma unroll
            for (int j = 0; j < NU / 2; j++) {
                const uint32_t vo = (uint32_t)(los + 64 * (j + 5 - role)) * (uint32_t)(NW * 4);
                ta[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_red, vo, 0u, 0);
                tb[j] = o_u32x2{0u, 0u};
                if (NW > 4) tb[j] = __builtin_amdgcn_raw_buffer_load_b64(rs_red, vo + 16u, 0u, 0);
            }
            uint32_t U[NU];
#pragma unroll
            for (int i = 0; i < NU; i++) U[i] = i < NW ? sacc[i] : 0u;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const bool lower = (los & sh) == 0;
                uint32_t X[NU], Y[NU];
#pragma unroll
                for (int i = 0; i < NU; i++) {
                    const uint32_t P = shx(U[i], sh);
                    X[i] = lower ? U[i] : P;      // the lower block's value: times x^s
                    Y[i] = lower ? P : U[i];
                }
#pragma unroll
                for (int i = NU - 1; i >= 0; i--) {
                    const uint32_t hi = X[i], lw = i > 0 ? X[i - 1] : 0u;
                    U[i] = (sh == 32 ? lw : ((hi << sh) | (lw >> (32 - sh)))) ^ Y[i];
                }
            }
            // V_w x^(64 (5 - w)) mod g(x): bit n = lo + 64 j of V_w (word 2 j + (lo >> 5): the same words in every lane now) selects the table's entry x^k mod g(x), k = n + 64 (5 - w)
            uint32_t part[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) part[i] = 0u;
#pragma unroll
            for (int j = 0; j < NU / 2; j++) {
                const uint32_t hsel = (uint32_t)((int32_t)((uint32_t)los << 26) >> 31);                     // all ones in the lanes 32 .. 63: they look at the odd word
                const uint32_t wsel = U[2 * j] ^ ((U[2 * j] ^ U[2 * j + 1]) & hsel);                  // (bitwise: a select of two array elements becomes a dynamically indexed array in private memory)
                const uint32_t mk = (uint32_t)((int32_t)(wsel << (31 - (los & 31))) >> 31);
                part[0] ^= ta[j].x & mk; part[1] ^= ta[j].y & mk; part[2] ^= ta[j].z & mk; part[3] ^= ta[j].w & mk;
                if (NW > 4) { part[NW > 4 ? 4 : 0] ^= tb[j].x & mk; part[NW > 4 ? 5 : 0] ^= tb[j].y & mk; }
            }
            // the wave's XOR over its lanes, then into the workgroup's words (zero at the start of every frame)
#pragma unroll
            for (int k = 0; k < NW; k++) {
                uint32_t x = part[k];
                for (int o = 32; o > 0; o >>= 1) x ^= shx(x, o);
                if (los == 0 && x) __hip_atomic_fetch_xor(s_misc + 16 + k, (int)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        const uint32_t t4o = SPA ? (uint32_t)to * 4u : t4;      // (sum-product kernel: re-formed from the opaque lane index, like the frame input's -- t4 kept alive across the layer loop costs the register it does not have)
        auto out_bar = [&]() __attribute__((always_inline)) {      // a barrier of the output phase: LDS traffic only (what the row-keeping waves hand over); the stores in flight are not waited for
#if W8_OUT_BAR_LGKM
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
            __syncthreads();
#endif
        };
        auto run_out_fast = [&](auto nw_c) __attribute__((always_inline)) {
            constexpr int NW = decltype(nw_c)::value;                  // words of the BCH remainder: 0 (bits socket alone: no verification), 4 or 6
            constexpr bool SYN = NW > 0;
            constexpr int IOS = SPA ? 8 : W8_IO;                         // rows per batch
            const const_u32 srow = rows + nl + ng + q;
            constexpr int NRP = w8_parked(MODE) ? ldpc_park_nr(MODE) : 0;
            // kind 0: LDS rows, 1: global rows, 2: the parked rows handed over through LDS positions 0 .. NR-1
            auto grp = [&](int kind, int l) __attribute__((always_inline)) -> uint32_t { return kind == 0 ? rows[l] : kind == 1 ? rows[nl + l] : srow[l]; };
            // A batch of B rows without a branch in it: the lanes without a check read a valid address (LDS) or beyond the slot (global rows: the access returns 0) and
            // store beyond their socket; the segment's remainder is its LAST B rows once more (`ldone`: rows below it were emitted by the batch before -- stored twice,
            // which the bits socket does not mind, and masked out of the BCH remainder).  With per-row tests every row was a basic block of its own: its row number, PRBS
            // word and buffer descriptor (spilled to a lane of a vector register) were fetched, and waited for, row by row.
            const uint32_t t4l = act ? t4o : 0u, t4g = act ? t4o : W8_OOB;
            auto body = [&](int kind, int l0, int ldone, auto b_c) __attribute__((always_inline)) {
                constexpr int B = decltype(b_c)::value;
                float v[B];
#pragma unroll
                for (int k = 0; k < B; k++) v[k] = kind == 1 ? gld(t4g, grow0 + (uint32_t)(l0 + k) * W8_ROW) : lld((uint32_t)(l0 + k) * W8_ROW + t4l);
                o_u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
                uint32_t pw = 0u;
                if (SYN) {      // the batch's records (lane l mod 16: row l) and this lane's PRBS bits of its rows
                    const uint32_t ks0 = (uint32_t)((kind == 0 ? 0 : kind == 1 ? nl_info : nl_info + ng_info) + l0);
                    r0 = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, rec_vo, ks0 * 32u, 0);
                    r1 = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, rec_vo + 16u, ks0 * 32u, 0);
                    pw = __builtin_amdgcn_raw_buffer_load_b32(rs_pl, pl_vo, ks0 * (uint32_t)(LDPC_AT_LANES * 4), 0);
                }
#pragma unroll
                for (int k = 0; k < B; k++) {
                    if (k > 0 && k % 8 == 0) __builtin_amdgcn_sched_barrier(0);
                    if (SYN) { emit_syn(k, v[k], l0 + k < ldone, r0, r1, pw, nw_c); continue; }
                    const uint32_t gk = grp(kind, l0 + k);
                    if (kind == 2) { if (gk != 0xFFFFFFFFu) emit_plain((int)gk, v[k]); }      // (an empty register slot: the plan of the DVB-S2 codes leaves none)
                    else emit_plain((int)gk, v[k]);
                }
            };
            auto seg = [&](int kind, int n) __attribute__((always_inline)) {
                int l0 = 0;
                for (; l0 + IOS <= n; l0 += IOS) body(kind, l0, 0, std::integral_constant<int, IOS>{});
                if (l0 < n) {
                    if (n >= IOS) body(kind, n - IOS, l0, std::integral_constant<int, IOS>{});
                    else for (; l0 < n; l0++) body(kind, l0, 0, std::integral_constant<int, 1>{});
                }
            };
            if (role >= 0) {
#if W8_OUT_GFIRST
                seg(1, ng_info);      // the global rows first: their loads are issued while nothing of this phase is in the wave's memory queue yet
                PROF_MARK_O(0);
                seg(0, nl_info);
                PROF_MARK_O(1);
#else
                seg(0, nl_info);
                PROF_MARK_O(1);
                seg(1, ng_info);
                PROF_MARK_O(0);
#endif
            }
            if (w8_parked(MODE)) {
                out_bar();      // every LDS row has been read
                out_bar();      // the parked rows are in positions 0 .. NR-1
                PROF_MARK_O(2);
                if (role >= 0) seg(2, NRP);
                PROF_MARK_O(3);
            }
            if constexpr (SYN) { if (role >= 0) fold_syn(nw_c); }
        };
        bool fast_out = false;
#if W8_FAST_OUT
        if (out_plain) { run_out_fast(std::integral_constant<int, 0>{}); fast_out = true; }
        else if (out_syn && syn6) { run_out_fast(std::integral_constant<int, 6>{}); fast_out = true; }
        else if (out_syn) { run_out_fast(std::integral_constant<int, 4>{}); fast_out = true; }
        else if (out_chain) { run_out(emit_chain); fast_out = true; }
#endif
        if (!fast_out) run_out(emit);
        PROF_MARK_O(4);
        if (first) s_misc[9] = p.cu_ctr ? (int)(atomicAdd(&p.cu_ctr[LDPC_FRAME_CTR], 1u) + gridDim.x) : qp + (int)gridDim.x;
        // the posterior image is reused by the next frame of this workgroup: every load of this phase has been consumed, the stores in flight touch the sockets only
        if (fast_out) out_bar(); else __syncthreads();
        PROF_MARK_O(5);
        if (out_syn && first) {
            int nz = 0;
#pragma unroll
            for (int k = 16; k < 22; k++) { nz |= s_misc[k]; s_misc[k] = 0; }      // (the next frame's first XOR is a frame input and a barrier away)
            p.bch_flag[fu] = nz ? 1 : 0;
            if (p.cwd_bch && !nz) p.cwd_bch[fu] = 1;
        }
        qp = SPA ? __builtin_amdgcn_readfirstlane(s_misc[9]) : s_misc[9];      // (uniform: the frame's base addresses stay on the scalar unit)
        PROF_MARK(7);
#ifdef LDPC_PHASE_PROF
        prof[9]++;                                            // frames this workgroup decoded
#endif
    }
#ifdef LDPC_PHASE_PROF
    if (lane == 0 && p.cu_ctr) {
        prof[11] = (uint32_t)(__builtin_amdgcn_s_memtime() - prof_t0);
        for (int i = 0; i < 12; i++) p.cu_ctr[LDPC_CU_CTR_WORDS + ((int)blockIdx.x * 8 + wave) * 12 + i] = role >= 0 ? prof[i] : 0u;
    }
#endif
}

template <int DEG, int MODE, int SPA = 0>
static hipError_t wg8_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_wg8_kernel<DEG, MODE, SPA>;
    static size_t configured_dev[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t &configured = configured_dev[dev & 63];
    const size_t lds = (size_t)pl.w8_lds_bytes;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    const int grid = p.n_frames < pl.grid_max ? p.n_frames : pl.grid_max;
    if (p.cu_ctr) { hipError_t e = hipMemsetAsync(p.cu_ctr, 0, LDPC_CU_CTR_WORDS * sizeof(uint32_t), s); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
#ifdef LDPC_PHASE_PROF
    {   // development aid: average ticks per phase over the working waves
        (void)hipStreamSynchronize(s);
        std::vector<uint32_t> hbuf((size_t)grid * 8 * 12);
        (void)hipMemcpy(hbuf.data(), p.cu_ctr + LDPC_CU_CTR_WORDS, hbuf.size() * 4, hipMemcpyDeviceToHost);
        double acc[12] = {0}; int nw = 0;
        for (int w = 0; w < grid * 8; w++) { if (!hbuf[(size_t)w * 12 + 11]) continue; nw++; for (int i = 0; i < 12; i++) acc[i] += hbuf[(size_t)w * 12 + i]; }
        #ifdef LDPC_PROF_OUT
        static const char *nm[12] = {"out: global rows", "out: LDS rows", "out: hand-over", "out: parked rows", "out: fold", "out: end barrier", "syndrome", "output (rest)", "input", "-frames", "-", "TOTAL"};
#else
        static const char *nm[12] = {"1a issue", "1b", "mid barrier", "pass 2", "replay", "end barrier", "syndrome", "output", "input", "-frames", "-", "TOTAL"};
#endif
        fprintf(stderr, "[ldpc phase prof] %d working waves\n", nw);
        {   // frames per workgroup and busy time: how evenly the work queue feeds the persistent grid
            std::vector<int> hist(64, 0); double tmin = 1e30, tmax = 0;
            for (int w = 0; w < grid * 8; w++) { const uint32_t *q = &hbuf[(size_t)w * 12]; if (!q[11] || (w & 7) >= 1) continue; hist[q[9] < 63 ? q[9] : 63]++; }
            for (int w = 0; w < grid * 8; w++) { const uint32_t *q = &hbuf[(size_t)w * 12]; if (!q[11]) continue; tmin = std::min(tmin, (double)q[11]); tmax = std::max(tmax, (double)q[11]); }
            fprintf(stderr, "  frames per workgroup (wave 0 of each):");
            for (int i = 0; i < 64; i++) if (hist[i]) fprintf(stderr, "  %d frames: %d WGs", i, hist[i]);
            fprintf(stderr, "\n  busy ticks per wave: min %.0f max %.0f\n", tmin, tmax);
        }
        for (int i = 0; i < 12; i++) if (nm[i][0] != '-') fprintf(stderr, "  %-18s %12.0f ticks/wave  %5.1f %%\n", nm[i], acc[i] / (nw ? nw : 1), 100.0 * acc[i] / (acc[11] > 0 ? acc[11] : 1));
    }
#endif
    return hipGetLastError();
}

template <int DEG, int MODE, int SPA = 0>
static int wg8_occ(const LdpcPlan &pl)
{
    auto kern = ldpc_wg8_kernel<DEG, MODE, SPA>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pl.w8_lds_bytes);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 512, (size_t)pl.w8_lds_bytes) != hipSuccess) nb = 1;
    if (nb > 2) nb = 2;          // the role assignment balances exactly two workgroups per CU
    if (const char *ev = getenv("DVBS2HIP_LDPC_BLOCKS_PER_CU")) { const int cap = atoi(ev); if (cap >= 1 && cap < nb) nb = cap; }
    return nb < 1 ? 1 : nb;
}

#define WG8_DISPATCH(FN, ...)                                                                                                              \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0>(__VA_ARGS__) : pl.fast_mode == 3 ? FN<27, 3>(__VA_ARGS__) : pl.fast_mode == 4 ? FN<27, 4>(__VA_ARGS__) : pl.fast_mode == 5 ? FN<27, 5>(__VA_ARGS__) : FN<27, 1>(__VA_ARGS__)) \
     : pl.fast_deg == 13 ? (pl.fast_mode == 0 ? FN<13, 0>(__VA_ARGS__) : FN<13, 1>(__VA_ARGS__))                                           \
                         : (pl.fast_mode == 0 ? FN<11, 0>(__VA_ARGS__) : FN<11, 1>(__VA_ARGS__)))

#define WG8_RULE_DISPATCH(RULE, FN, ...)                                                                                                                   \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0, RULE>(__VA_ARGS__) : pl.fast_mode == 3 ? FN<27, 3, RULE>(__VA_ARGS__) : pl.fast_mode == 4 ? FN<27, 4, RULE>(__VA_ARGS__) : FN<27, 1, RULE>(__VA_ARGS__)) \
     : pl.fast_deg == 13 ? (pl.fast_mode == 0 ? FN<13, 0, RULE>(__VA_ARGS__) : FN<13, 1, RULE>(__VA_ARGS__))                                                 \
                         : (pl.fast_mode == 0 ? FN<11, 0, RULE>(__VA_ARGS__) : FN<11, 1, RULE>(__VA_ARGS__)))
#define WG8_ANY_DISPATCH(FN, ...)                                                                                  \
    (pl.spa_rule == 2 ? WG8_RULE_DISPATCH(2, FN, __VA_ARGS__) : pl.spa_rule == 3 ? WG8_RULE_DISPATCH(3, FN, __VA_ARGS__) \
     : pl.spa ? WG8_RULE_DISPATCH(1, FN, __VA_ARGS__) : WG8_DISPATCH(FN, __VA_ARGS__))

int ldpc_wg8_blocks_per_cu(const LdpcPlan &pl) { return WG8_ANY_DISPATCH(wg8_occ, pl); }

hipError_t ldpc_wg8_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    p.cu_ctr = pl.d_cu_ctr;
    p.w8.tab = pl.d_w8_tab; p.w8.rows = pl.d_w8_rows; p.w8.atab = pl.d_w8_atab;
    p.w8.st_base = pl.w8_st_base; p.w8.lds_junk = pl.w8_lds_junk; p.w8.lds_bytes = pl.w8_lds_bytes; p.w8.pad = pl.fast_pad ? 1 : 0;
    p.w8.nl_info = pl.w8_nl_info; p.w8.nl = pl.w8_nl; p.w8.ng_info = pl.w8_ng_info; p.w8.ng = pl.w8_ng;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    p.gwork_words = pl.w8_gwork_words;
    p.spa_cap = pl.spa_rule == 3 ? LDPC_SPA_CAP : INFINITY;
    return WG8_ANY_DISPATCH(wg8_inst, pl, p, s);
}

}  // namespace dvbs2
