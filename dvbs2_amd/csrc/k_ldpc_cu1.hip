// a1, mode 6 -- layered NMS / MS, ONE frame per 16-wave workgroup = per CU, the WHOLE posterior image on chip.
//
// k_ldpc_wg8.hip (modes 3-5) runs two frames per CU, one lane per check, and keeps 94 of a normal frame's 180 bit-group rows on chip; the other 86 cross the
// fabric behind L2 twice per iteration (7.7 MB per frame), and a wave's instruction stream of ~770 instructions per layer leaves the CU's two frames waiting
// for each other's issue slots and for those rows.  Here:
//   * TWO LANES PER CHECK, in different waves: wave role (half, block) -- half A's lane t takes slots 0 .. 13 of check t of the layer (the plan puts every
//     duplicate edge there), half B's lane t slots 14 .. 26 (p_c and p_{c-1} last: the parity chain travels in one of B's registers as in modes 3-5).  Each half
//     folds its slots into {min1, min2, parity of the signs}, the two halves swap those through 8 bytes of LDS per half-check behind the layer's first barrier
//     (which every layer with duplicate edges needed anyway: all reads before the first write) and merge them: min1 = min(a1, b1), min2 = min(max(a1, b1), a2, b2)
//     -- the two smallest of the union, the same two fp32 values the one-lane scan finds -- so the messages, the posteriors and the packed state are bit for bit
//     those of the oracle's ORC_SCHED_QC.  A wave's stream per layer is half as long, and 12 working waves (3 per SIMD) share one frame's layer.
//   * 108 rows in LDS positions + 72 rows PARKED in the registers of the workgroup's four row-keeping waves (two groups of two waves, 36 rows each, 3 VGPRs per
//     row and lane, swapped with their partner row's position by the static cyclic schedule of k_ldpc.hip, plan_parked): 180 of 180 rows on chip, every slot an
//     LDS access, no posterior ever leaves the CU.  What still goes to memory: the packed check state (16 bytes per check and layer: {c1, c2, signs | position of
//     the minimum for half A, the same for half B}; 115 KB per frame, L2-resident) and the frame's LLRs in / decisions out.
//   * Same layer tables as k_ldpc_wg8.hip (one dword per slot: byte shift | LDS byte offset of the row's position in THAT layer), same stopping rule, outputs and
//     work queue (frames handed out through a counter).
// Same schedule and arithmetic as the oracle (bit-exact): tests/test_ldpc_gpu.py, tests/test_golden_gpu.py.
#include "dvbs2hip_internal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace dvbs2 {

typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) int lds_int;
typedef const __attribute__((address_space(4))) uint32_t *const_u32;

constexpr int C1_ROW = LDPC_Z * 4;           // bytes per bit-group row
constexpr int C1_IO = 16;                    // independent loads per lane in flight during frame I/O
constexpr int C1_HA = LDPC_CU1_HA;           // slots of half A
#ifndef LDPC_CU1_HA_SPA_V
#define LDPC_CU1_HA_SPA_V 13
#endif
constexpr int C1_HA_SPA = LDPC_CU1_HA_SPA_V;    // ... in the sum-product kernel (half A also keeps the duplicate edges' deltas: 13 + 14 balances the registers)
constexpr int C1_WAVES = 16, C1_THREADS = C1_WAVES * 64;

// development knobs (tools/build_variant_tus.sh): the row-keeping waves swap with ds_wrxchg_rtn (0) or with a read and a write per word (1); they swap right behind the
// end-of-layer barrier (0) or behind the layer's first barrier (1), when the working waves' loads are out of the way
#ifndef C1_SPA_BS        // sum-product layer: suffix values kept for every C1_SPA_BS-th slot
#define C1_SPA_BS 2
#endif
#ifndef C1_SPA_KEEPW     // sum-product layer: a slot's LDS address is kept from its load to its store (1) or formed twice (0)
#define C1_SPA_KEEPW 1
#endif
#ifndef C1_SPA_FENCE
#define C1_SPA_FENCE 2
#endif
#ifndef C1_KEEP_RW
#define C1_KEEP_RW 0
#endif
#ifndef C1_KEEP_LATE
#define C1_KEEP_LATE 0
#endif
#ifdef LDPC_PHASE_PROF
#define C1_MARK(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof[i] += (uint32_t)(n_ - pt_); pt_ = n_; } while (0)
#else
#define C1_MARK(i)
#endif

__device__ __forceinline__ lds_float *c1_lds(uint32_t a) { return (lds_float *)(size_t)a; }

__device__ __forceinline__ float c1_and_or(uint32_t a, uint32_t m_sgpr, float b)      // (a & m) | b in one VALU operation
{
    float r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m_sgpr), "v"(b));
    return r;
}

// message on local slot j of a half with NS slots from the packed per-check state: magnitude c1 at the recorded minimum, c2 elsewhere; sign = bit (NS-1-j) of pk
template <int NS>
__device__ __forceinline__ float c1_unpack(float c1, float c2, uint32_t pk, uint32_t j, uint32_t sb)
{
    const float mag = ((pk >> 27) == j) ? c1 : c2;
    return c1_and_or(pk << ((32u - NS) + j), sb, mag);
}

// ---- the row-keeping waves: group `grp` (two waves, 120 of their 128 lanes hold elements e, e + 120, e + 240 of NRG rows) follows the working waves barrier for
// barrier and swaps rows with LDS positions by the plan's table (w8_park_server of k_ldpc_wg8.hip, for a workgroup with two such groups and the barrier
// sequence of this kernel's layer: one barrier always, then the duplicate-edge barriers, then the end barrier)
template <int NRG, int SPA>      // SPA: 0 min-sum, 1 sum-product (exact, per-check scale: two exchanges per layer), 3 sum-product with AFF3CT's cap (no scale: one exchange)
__device__ __forceinline__ void cu1_keeper(const LdpcKParams &p, lds_int *const s_misc, const int grp, const int sidx, const int lane, const bool vote_writer
#ifdef LDPC_PHASE_PROF
                                           , uint32_t *prof, unsigned long long &pt_
#endif
                                           )
{
    const int q = p.q;
    constexpr int NRT = 2 * NRG;
    const const_u32 tab = (const_u32)p.w8.tab;
    // the swaps of layer r as a 64-bit mask per group (bit k: slot k <-> LDS position grp * NRG + k; the plan gives pair k position k), behind the layer tables
    // and the [q][NRT] byte table k_ldpc_wg8.hip's modes read
    const const_u32 swm = tab + q * LDPC_FAST_STRIDE + q * NRT + grp * 2;
    const const_u32 srow = (const_u32)p.w8.rows + p.w8.nl + p.w8.ng + q + grp * NRG;        // bit-group in this group's slot k at the start of an iteration
    const int el = sidx * 64 + lane;
    const bool on = el < LDPC_Z / 3;
    const uint32_t a0 = (uint32_t)el * 4u + (uint32_t)(grp * NRG) * (uint32_t)C1_ROW;       // this lane's first word of the group's first position
    constexpr uint32_t A1 = LDPC_Z / 3 * 4u, A2 = 2u * A1;
    auto lst = [&](uint32_t a, float v) { *c1_lds(a) = v; };
    auto lxc = [&](uint32_t a, float v) -> float { return __hip_atomic_exchange(c1_lds(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    float R[NRG][3];
    int nvote = 0;
    // The barriers INSIDE a layer are taken without waiting for the exchanges in flight (a plain s_barrier: the working waves do not look at the rows being
    // swapped before the layer's END barrier, which is a full __syncthreads); a barrier that drained them made the twelve working waves wait for this one.
    auto bar_nowait = [&]() { __builtin_amdgcn_s_barrier(); };
    auto vote0 = [&]() -> bool {                    // the working waves' vote, with nothing to report
        lds_int *const w = s_misc + 20;
        const int k = nvote % 3;
        nvote++;
        if (vote_writer && lane == 0) w[(k + 1) % 3] = 0;
        __syncthreads();
        return w[k] != 0;
    };
    auto moves = [&](int r) {                       // the swaps of layer r: every register index and every LDS offset is a compile-time constant, the test is scalar
        const uint32_t mlo = swm[r * 4], mhi = swm[r * 4 + 1];
        if (on)
#pragma unroll
        for (int k = 0; k < NRG; k++) {
            const bool sw = k < 32 ? ((mlo >> k) & 1u) != 0u : ((mhi >> (k - 32)) & 1u) != 0u;
            if (sw) {
                const uint32_t b = a0 + (uint32_t)k * (uint32_t)C1_ROW;
#if C1_KEEP_RW
                const float n0 = *c1_lds(b), n1 = *c1_lds(b + A1), n2 = *c1_lds(b + A2);
                lst(b, R[k][0]); lst(b + A1, R[k][1]); lst(b + A2, R[k][2]);
                R[k][0] = n0; R[k][1] = n1; R[k][2] = n2;
#else
                R[k][0] = lxc(b, R[k][0]); R[k][1] = lxc(b + A1, R[k][1]); R[k][2] = lxc(b + A2, R[k][2]);
#endif
            }
        }
    };
    auto layer_barriers = [&](int r) {              // the barriers of one decoding layer, as the working waves take them
        const const_u32 T = tab + r * LDPC_FAST_STRIDE;
        const uint32_t cinfo = T[28];
        const int ncf = (int)(cinfo & 0xFFu);
        bar_nowait();                               // the halves' partial minima are in the exchange area; every read of the layer precedes its writes
        if (SPA == 1) bar_nowait();                 // sum-product layer with the per-check scale: the halves' products are in the exchange area (the capped rule needs one exchange)
#if C1_KEEP_LATE
        moves(r);
#endif
        if (ncf > 0) {
            bar_nowait();                           // the primary writes are in place
            uint32_t prev_lvl = 1u;
            for (int i = (ncf > 1 && ((cinfo >> 21) & 3u) == 1u) ? 2 : 1; i < ncf; i++) {
                const uint32_t lvl = T[48 + i] >> 8;
                if (lvl != prev_lvl) { bar_nowait(); prev_lvl = lvl; }
            }
        }
        C1_MARK(0);
        __syncthreads();                            // end of the layer: the swapped rows are in place
        C1_MARK(1);
    };
    const bool es = p.early_stop != 0;
    for (int qp = blockIdx.x; qp < p.n_frames; ) {
        const int f = p.order ? (int)p.order[qp] : qp;      // (queue position -> frame: LdpcKParams::order, k_ldpc_wg8.hip)
        const float *Y = p.llr + (size_t)f * p.N;
        const int elc = on ? el : 0;
#pragma unroll
        for (int k = 0; k < NRG; k++) {
            const uint32_t g = srow[k];
            if (g != 0xFFFFFFFFu && (int)g >= p.n_info) {      // a parity group: element e is bit K + q e + (g - n_info)
                const float *Yg = Y + p.K + ((int)g - p.n_info);
                R[k][0] = Yg[q * elc]; R[k][1] = Yg[q * (elc + LDPC_Z / 3)]; R[k][2] = Yg[q * (elc + 2 * (LDPC_Z / 3))];
                continue;
            }
            const float *Yg = Y + (g == 0xFFFFFFFFu ? 0u : g) * (uint32_t)LDPC_Z;
            R[k][0] = __builtin_nontemporal_load(&Yg[elc]); R[k][1] = __builtin_nontemporal_load(&Yg[elc + LDPC_Z / 3]); R[k][2] = __builtin_nontemporal_load(&Yg[elc + 2 * (LDPC_Z / 3)]);
        }
        __syncthreads();                            // the image is in place
        C1_MARK(8);
        // ph 0: layer r of an iteration; 1: layer r of the syndrome sweep; 2: catching up with the schedule after a sweep that stopped early
        int ph = 0, r = 0, it = 0;
        bool ok = false, fin = false;
        while (!fin) {
            bool mv = true;
            if (ph == 1 && es && r == 0 && vote0()) { ok = false; mv = false; ph = 3; }       // stopping rule: the vote on layer 0 comes before anything moves
            if (mv && !(C1_KEEP_LATE && ph == 0)) moves(r);
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
        C1_MARK(6);
        // outputs: the parked rows go through LDS positions grp * NRG .. once the working waves have read the LDS rows
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NRG; k++) if (on) { const uint32_t b = a0 + (uint32_t)k * (uint32_t)C1_ROW; lst(b, R[k][0]); lst(b + A1, R[k][1]); lst(b + A2, R[k][2]); }
        __syncthreads();
        __syncthreads();                            // the image is reused by the next frame
        qp = s_misc[19];
        C1_MARK(7);
#ifdef LDPC_PHASE_PROF
        prof[9]++;
#endif
    }
}

// ---- one half-check's lanes.  HALF 0: slots 0 .. 13 (duplicate edges among the first ones, replayed by these lanes); HALF 1: slots 14 .. 26 (25 = p_c, 26 = p_{c-1})
template <int DEG, int HALF, int SPA, int HA>
__device__ __forceinline__ void cu1_work(const LdpcKParams &p, lds_int *const s_misc, const int cb, const int lane, const int wave, const bool vote_writer, const bool first_wave
#ifdef LDPC_PHASE_PROF
                                         , uint32_t *prof, unsigned long long &pt_
#endif
                                         )
{
    constexpr int NS = HALF == 0 ? HA : DEG - HA, J0 = HALF == 0 ? 0 : HA;      // local slot i is the layer's slot J0 + i
    constexpr bool FWD = HALF == 1;                  // parity chain forwarded in a register: p_c at local slot NS-2, p_{c-1} at NS-1
    const int t = cb * 64 + lane;
    const bool act = t < LDPC_Z;
    const uint32_t t4 = (uint32_t)t * 4u;
    const int q = p.q;
    float *gwork = p.gwork + (size_t)blockIdx.x * p.gwork_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gwork, 0, p.gwork_words * 4, 0x00020000);
    const const_u32 tab = (const_u32)p.w8.tab;
    const const_u32 rows = (const_u32)p.w8.rows;
    const uint32_t ljunk = p.w8.lds_junk;
    const uint32_t xch = ljunk + (uint32_t)C1_ROW;                    // exchange area: [half][360] x 8 bytes
    const uint32_t xmine = xch + (uint32_t)HALF * (LDPC_Z * 8u) + (uint32_t)t * 8u, xother = xch + (uint32_t)(1 - HALF) * (LDPC_Z * 8u) + (uint32_t)t * 8u;
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));
    auto wide_off = [&](uint32_t voff, uint32_t soff) __attribute__((always_inline)) -> uint32_t { uint32_t o = voff + soff; asm volatile("" : "+v"(o)); return o; };      // (see k_ldpc_wg8.hip: stores of more than 8 bytes)
    auto lld = [&](uint32_t a) __attribute__((always_inline)) -> float { return *c1_lds(a); };
    auto lst = [&](uint32_t a, float v) __attribute__((always_inline)) { *c1_lds(a) = v; };
    const int nl_info = p.w8.nl_info, nl = p.w8.nl;
    int nvote = 0;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

    for (int qp = blockIdx.x; qp < p.n_frames; ) {
        const int f = p.order ? (int)p.order[qp] : qp;
        // ---- channel LLRs -> the LDS positions that hold a row at the start of an iteration (the row-keeping waves load theirs themselves); the two halves take
        //      alternate batches of rows
        const float *Y = p.llr + (size_t)f * p.N;
        if (act) {
            for (int l0 = HALF * C1_IO; l0 < nl_info; l0 += 2 * C1_IO) {
                float v[C1_IO];
#pragma unroll
                for (int k = 0; k < C1_IO; k++) { const int g = (int)rows[l0 + k < nl_info ? l0 + k : nl_info - 1]; v[k] = __builtin_nontemporal_load(&Y[(g < p.n_info ? g : 0) * LDPC_Z + t]); }
#pragma unroll
                for (int k = 0; k < C1_IO; k++) if (l0 + k < nl_info && (int)rows[l0 + k < nl_info ? l0 + k : 0] < p.n_info) lst((uint32_t)(l0 + k) * C1_ROW + t4, v[k]);      // (a parity group's position: filled by the scatter below)
            }
        }
        {
            // parity part: bit K + q t + r belongs to row n_info + r, element t.  Each wave reads consecutive LLRs of its 64 checks coalesced (half A the
            // first half of the q * 64 values, half B the rest) and scatters them into the parity rows' positions.
            const int tl0 = cb * 64, cnt = (LDPC_Z - tl0 < 64 ? LDPC_Z - tl0 : 64) * q;
            const float *Yp = Y + p.K + tl0 * q;
            const const_u32 prow = rows + nl + p.w8.ng;
            const int dq = 64 / q, dr = 64 - dq * q;
            const int ih = (q + 1) / 2, i_lo = HALF == 0 ? 0 : ih, i_hi = HALF == 0 ? ih : q;
            int tl = (i_lo * 64 + lane) / q, r = (i_lo * 64 + lane) - tl * q;
            constexpr int PIO = 10;
            for (int i0 = i_lo; i0 < i_hi; i0 += PIO) {
                float v[PIO];
#pragma unroll
                for (int k = 0; k < PIO; k++) { const int e = (i0 + k) * 64 + lane; v[k] = (i0 + k < i_hi && e < cnt) ? Yp[e] : 0.f; }
#pragma unroll
                for (int k = 0; k < PIO; k++) {
                    if (i0 + k < i_hi) {
                        if ((i0 + k) * 64 + lane < cnt && prow[r] != 0xFFFFFFFFu) lst(prow[r] + (uint32_t)(tl0 + tl) * 4u, v[k]);      // (0xFFFFFFFF: that group starts in a register slot)
                        tl += dq; r += dr;
                        if (r >= q) { r -= q; tl++; }
                    }
                }
            }
        }
        if (p.packed && first_wave && lane == 0 && (p.K & 31)) p.packed[(size_t)f * ((p.K + 31) / 32) + p.K / 32] = 0u;     // the bits behind K in the last word
        __syncthreads();
        C1_MARK(8);

        int it = 0;
        bool ok = false;
        float nx1 = 0.f, nx2 = 0.f, nxk = 0.f;           // packed state of the next layer (prefetched): c1, c2, this half's signs | position
        float pfw = 0.f;                                 // HALF 1: posterior of parity bit q t + r after layer r, on its way to layer r + 1
        float onx[SPA ? NS : 1];                         // SPA: old messages of the NEXT layer, requested while this layer's stores drain
#pragma unroll
        for (int j = 0; j < (SPA ? NS : 1); j++) onx[j] = 0.f;
        uint32_t TE[32];                                 // layer table of the NEXT layer, fetched under the end-of-layer barrier
#pragma unroll
        for (int j = 0; j < 32; j++) TE[j] = SPA ? 0u : tab[j];
        // sum-product kernel: the next layer's table travels in ONE vector register (lane j holds entry j, read back with v_readlane where an entry is needed): carried as 32
        // scalars across the layer's barriers it does not fit the scalar file beside this layer's, and the compiler parks it in 32 VECTOR registers (and spills others)
        uint32_t tev = SPA ? p.w8.tab[lane & 31] : 0u;
        while (it < p.n_ite) {
            for (int r = 0; r < q; r++) {
                const const_u32 T = tab + r * LDPC_FAST_STRIDE;
                uint32_t E[NS];
#pragma unroll
                for (int j = 0; j < NS; j++) E[j] = SPA ? (uint32_t)__builtin_amdgcn_readlane((int)tev, J0 + j) : TE[J0 + j];
                const uint32_t prim = (SPA ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 27) : TE[27]) >> J0, cinfo = SPA ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 28) : TE[28],
                               ce0 = SPA ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 29) : TE[29], ce1 = SPA ? (uint32_t)__builtin_amdgcn_readlane((int)tev, 30) : TE[30];
                const int ncf = (int)(cinfo & 0xFFu);
                const bool mask0 = HALF == 1 && (r == 0) && (t == 0);        // p_{c-1} of check 0 does not exist
                if constexpr (SPA) {
                    // ================= sum-product layer, two lanes per check (round 5) =================
                    // The check node in the complement-product domain (k_ldpc_wg8.hip, SPA layer: |out_j| = ln((2 - Q_j) / Q_j), Q_j = 1 - prod_{i != j} (1 - u_i),
                    // u_i = 2 / (e^a_i + 1), carried as Q' = 2^s2 Q with s2 from the check's SECOND smallest magnitude) splits cleanly over the two halves of a check:
                    // Q_ab = Q_a + Q_b (1 - kap Q_a) is associative and commutative, so a half forms the prefix / suffix values of its own 13 / 14 slots and its own
                    // total, and an edge's Q_j is comb(comb(prefix_j, suffix_j), the OTHER half's total).  Two exchanges through the 8 bytes per half-check of the
                    // min-sum form: {min1 | parity of the signs, min2} behind the loads (the scale s2 and the overflow rule need the two smallest magnitudes of the
                    // WHOLE check; this barrier is also "every read of the layer precedes its writes"), then the half's total -- written into the OTHER half's
                    // slot, whose content this lane has just consumed, so the area needs no second copy.  The c->v messages, one fp32 per edge, live in the
                    // workgroup's global slot as [layer][half][group of 4 slots][360][4]: 778 KB per frame, 199 MB for the 256 workgroups of a launch -- inside
                    // the 256 MB Infinity Cache, which the 512 x 778 KB of the two-frames-per-CU kernel are not (there the messages are an HBM stream).
                    constexpr int NG4 = NS / 4, NT = NS % 4;
                    constexpr uint32_t HB = HALF == 0 ? 0u : (uint32_t)((HA / 4) * (C1_ROW * 4) + (HA % 4) * C1_ROW), LB = (uint32_t)(DEG * C1_ROW);
                    typedef uint32_t m_u32x4 __attribute__((ext_vector_type(4)));
                    typedef uint32_t m_u32x2 __attribute__((ext_vector_type(2)));
                    typedef uint32_t m_u32x3 __attribute__((ext_vector_type(3)));
                    auto msg_ld = [&](float *dst, uint32_t lbase) __attribute__((always_inline)) {
#pragma unroll
                        for (int g = 0; g < NG4; g++) {
                            const m_u32x4 q4 = __builtin_amdgcn_raw_buffer_load_b128(rs, t4 * 4u, lbase + HB + (uint32_t)g * (C1_ROW * 4u), 0);
                            dst[4 * g] = __uint_as_float(q4.x); dst[4 * g + 1] = __uint_as_float(q4.y); dst[4 * g + 2] = __uint_as_float(q4.z); dst[4 * g + 3] = __uint_as_float(q4.w);
                        }
                        const uint32_t tb = lbase + HB + (uint32_t)NG4 * (C1_ROW * 4u);
                        if (NT == 3) { const m_u32x3 q3 = __builtin_amdgcn_raw_buffer_load_b96(rs, t4 * 3u, tb, 0); dst[4 * NG4] = __uint_as_float(q3.x); dst[4 * NG4 + 1] = __uint_as_float(q3.y); dst[(NT == 3 ? 4 * NG4 + 2 : 0)] = __uint_as_float(q3.z); }
                        if (NT == 2) { const m_u32x2 q2 = __builtin_amdgcn_raw_buffer_load_b64(rs, t4 * 2u, tb, 0); dst[4 * NG4] = __uint_as_float(q2.x); dst[(NT >= 2 ? 4 * NG4 + 1 : 0)] = __uint_as_float(q2.y); }
                        if (NT == 1) dst[4 * NG4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t4, tb, 0));
                    };
                    uint32_t MAGM = 0x7FFFFFFFu;
                    asm volatile("" : "+s"(MAGM));
                    // registers: the suffix values are kept for every second slot and rebuilt from there on the way forward; the circulant offsets are formed twice (loads,
                    // stores) through an opaque copy of the lane's offset so that the compiler does not hold them across the arithmetic (as in k_ldpc_wg8.hip's SPA layer)
                    constexpr int BS = C1_SPA_BS, NB = (NS + BS - 1) / BS;
                    float x[NS], u[NS], Bs[NB], od[LDPC_SPA_MAXC];
                    uint32_t t4s = t4;
                    uint32_t wk[C1_SPA_KEEPW ? NS : 1];
                    auto woff = [&](int j, uint32_t tt) __attribute__((always_inline)) { const uint32_t d = tt - (E[j] & 0x7FFu); return min(d, d + (uint32_t)C1_ROW) + ((E[j] >> 11) & 0x3FFFFu); };
                    float mn1 = INFINITY, mn2 = INFINITY, kap = 1.f, cln = 0.f, key = 0.f, Qown = 0.f;
                    uint32_t sx = 0u;
                    auto comb = [&](float a, float b) __attribute__((always_inline)) { return __builtin_fmaf(b, __builtin_fmaf(-kap, a, 1.f), a); };      // Q'_ab
#pragma unroll
                    for (int i = 0; i < LDPC_SPA_MAXC; i++) od[i] = 0.f;
                    if (act) {
#pragma unroll
                        for (int j = 0; j < NS; j++) {
#if C1_SPA_KEEPW
                            wk[j] = woff(j, t4);
                            if (FWD && j == NS - 1 && r > 0) x[j] = pfw; else x[j] = lld(wk[j]);
#else
                            if (FWD && j == NS - 1 && r > 0) x[j] = pfw;         // p_{c-1}: handed over by layer r - 1
                            else x[j] = lld(woff(j, t4));
#endif
                        }
                        if (HALF == 0) {
#pragma unroll
                            for (int j = 0; j < LDPC_SPA_MAXC; j++) od[j] = onx[j];
                        }
#pragma unroll
                        for (int j = 0; j < NS; j++) x[j] = x[j] - onx[j];      // zeros in the first iteration
                        if (HALF == 1 && mask0) x[NS - 1] = INFINITY;
                        if constexpr (SPA == 3) {
                            // (round 6) `--dec-implem SPA`: the clip at 16.64 on the way out makes the per-check scale, the minima it is chosen from and the overflow rule
                            // unnecessary (k_ldpc_wg8.hip, SPA = 3) -- and with them the halves' FIRST exchange: this half's product and sign parity go out together
#pragma unroll
                            for (int j = 0; j < NS; j++) {
                                sx ^= __float_as_uint(x[j]);
                                u[j] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(__builtin_fmaf(fabsf(x[j]), 1.44269504088896341f, -1.f)) + 0.5f);
                            }
                            float b = 0.f;
#pragma unroll
                            for (int j = NS - 1; j >= 0; j--) { if (j % BS == BS - 1 || j == NS - 1) Bs[j / BS] = b; b = __builtin_fmaf(u[j], 1.f - b, b); }
                            Qown = b;
                            *c1_lds(xmine) = __uint_as_float(__float_as_uint(Qown) | (sx & 0x80000000u));      // (0 <= Qown <= 1: bit 31 is free for the sign parity)
                        } else {
#pragma unroll
                        for (int j = 0; j < NS; j++) {
                            const float a = fabsf(x[j]);
                            mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                            mn1 = fminf(mn1, a);
                            sx ^= __float_as_uint(x[j]);
                        }
                        f32x2 mine;
                        mine.x = __uint_as_float(__float_as_uint(mn1) | (sx & 0x80000000u)); mine.y = mn2;      // (+inf | sign bit: the magnitude bits stay those of +inf)
                        *(lds_f32x2 *)(size_t)xmine = mine;
                        }
                    }
                    C1_MARK(1);
                    __syncthreads();         // the halves' minima (SPA = 3: products) are in place; every read of the layer precedes its writes
                    C1_MARK(2);
                    uint32_t SXT = 0u;
                    float Aoth = 0.f;
                    if constexpr (SPA == 3) {
                        if (act) { const float o = *c1_lds(xother); SXT = (sx ^ __float_as_uint(o)) & 0x80000000u; Aoth = fabsf(o); }
                    } else {
                    if (act) {
                        const f32x2 oth = *(lds_f32x2 *)(size_t)xother;
                        const float o1 = fabsf(oth.x), o2 = oth.y;
                        SXT = (sx ^ __float_as_uint(oth.x)) & 0x80000000u;      // the parity of ALL the check's signs, in bit 31
                        const float hi = fmaxf(mn1, o1);
                        mn1 = fminf(mn1, o1);
                        mn2 = fminf(fminf(mn2, o2), hi);
                        const float s2 = fmaxf(0.f, (mn2 - 16.f) * 1.44269504088896341f);
                        kap = __builtin_amdgcn_exp2f(-s2);
                        cln = s2 * 0.693147180559945309f;
                        const float s2p1 = s2 + 1.f, hk = 0.5f * kap;
                        key = (mn2 - mn1 > 60.f) ? mn1 : __builtin_nanf("");
#pragma unroll
                        for (int j = 0; j < NS; j++) {
                            const float ea = __builtin_fmaf(fabsf(x[j]), 1.44269504088896341f, -s2p1);
                            u[j] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(ea) + hk);
                        }
                        float b = 0.f;      // B_j = Q' of this half's slots behind j; Bs[k] = B_{BS k + BS - 1} (clipped to the last slot)
#pragma unroll
                        for (int j = NS - 1; j >= 0; j--) { if (j % BS == BS - 1 || j == NS - 1) Bs[j / BS] = b; b = comb(b, u[j]); }
                        Qown = b;
                        *c1_lds(xother) = Qown;      // into the other half's slot (its content has been consumed above); that half reads its own slot behind the barrier
                    }
                    __syncthreads();         // the halves' totals are in place
                    }
                    if (act) {
                        // the prefix recursion STARTS from the other half's total: A_j then holds "the other half and this half's slots before j", and Q_j = comb(A_j, B_j)
                        // needs no third operand (two fused multiply-adds per slot less than comb(comb(A_j, B_j), Q_other))
                        float A = SPA == 3 ? Aoth : *c1_lds(xmine);
                        const bool anykey = __ballot(key == key) != 0ull;      // some check of this wave is in the overflow case (min2 - min1 > 60): rare, and wave-uniform
                        float mq[4] = {0.f, 0.f, 0.f, 0.f};
                        const uint32_t mrow = (uint32_t)r * LB;
                        asm volatile("" : "+v"(t4s));
#pragma unroll
                        for (int j = 0; j < NS; j++) {
                            // (a scheduling fence every C1_SPA_FENCE slots: left alone the scheduler interleaves all 13 / 14 slots' reciprocal / logarithm chains and needs
                            // ~30 registers more than the kernel has)
                            if (C1_SPA_FENCE > 0 && j > 0 && j % C1_SPA_FENCE == 0) __builtin_amdgcn_sched_barrier(0);
                            const float wA = __builtin_fmaf(-kap, A, 1.f);
                            float Bj;
                            {
                                const int js = (j / BS) * BS + BS - 1 < NS - 1 ? (j / BS) * BS + BS - 1 : NS - 1;      // the kept slot at or behind j
                                Bj = Bs[j / BS];
#pragma unroll
                                for (int i = js; i > j; i--) Bj = comb(Bj, u[i]);
                            }
                            const float Q = __builtin_fmaf(Bj, wA, A);                    // every slot of the check but j
                            const float lg = __builtin_amdgcn_logf(__builtin_fmaf(2.f, __builtin_amdgcn_rcpf(Q), -kap));
                            float o = SPA == 3 ? lg * 0.693147180559945309f : __builtin_fmaf(lg, 0.693147180559945309f, cln);
                            if (SPA != 3 && anykey) { asm volatile("" ::: "memory"); o = __builtin_islessgreater(fabsf(x[j]), key) ? mn1 : o; }      // (the empty asm keeps the compiler from turning the wave-uniform branch back into two selects per slot)      // (ordered "not equal": false against the NaN that stands for "no overflow")
                            o = fminf(o, p.spa_cap);      // (`--dec-implem SPA`: the cap of AFF3CT's tanh-product rule; SPA_EXACT: +inf)
                            float nw;
                            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(nw) : "s"(MAGM), "v"(o), "v"(SXT ^ __float_as_uint(x[j])));
                            A = __builtin_fmaf(u[j], wA, A);
                            uint32_t a = C1_SPA_KEEPW ? wk[C1_SPA_KEEPW ? j : 0] : woff(j, t4s);
                            if (HALF == 1 && j == NS - 1 && mask0) a = ljunk + t4s;
                            if (FWD && j == NS - 2 && r + 1 < q) pfw = x[j] + nw;                    // p_c: kept for layer r + 1
                            else if (HALF == 0 && j < ldpc_w8_kd(DEG)) { if (((prim >> j) & 1u) != 0u) lst(a, x[j] + nw); }      // a duplicate edge's plain store is left out
                            else lst(a, x[j] + nw);
                            mq[j & 3] = nw;
                            if ((j & 3) == 3)
                                __builtin_amdgcn_raw_buffer_store_b128(m_u32x4{__float_as_uint(mq[0]), __float_as_uint(mq[1]), __float_as_uint(mq[2]), __float_as_uint(mq[3])}, rs,
                                                                       wide_off(t4s * 4u, mrow + HB + (uint32_t)(j >> 2) * (C1_ROW * 4u)), 0u, 0);
                            else if (j == NS - 1) {
                                const uint32_t tb = mrow + HB + (uint32_t)NG4 * (C1_ROW * 4u);
                                if (NT == 3) __builtin_amdgcn_raw_buffer_store_b96(m_u32x3{__float_as_uint(mq[0]), __float_as_uint(mq[1]), __float_as_uint(mq[2])}, rs, wide_off(t4s * 3u, tb), 0u, 0);
                                if (NT == 2) __builtin_amdgcn_raw_buffer_store_b64(m_u32x2{__float_as_uint(mq[0]), __float_as_uint(mq[1])}, rs, t4s * 2u, tb, 0);
                                if (NT == 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mq[0]), rs, t4s, tb, 0);
                            }
                            if (HALF == 0 && j < LDPC_SPA_MAXC) od[j] = nw - od[j];
                        }
                        // the next layer's messages: requested now, they travel while this layer's stores drain at the barriers
                        __builtin_amdgcn_sched_barrier(0);
                        if ((it > 0 || r + 1 == q) && !(r + 1 == q && it + 1 >= p.n_ite)) msg_ld(onx, (uint32_t)(r + 1 < q ? r + 1 : 0) * LB);
                        else {
#pragma unroll
                            for (int j = 0; j < NS; j++) onx[j] = 0.f;
                        }
                    }
                    tev = p.w8.tab[(r + 1 < q ? r + 1 : 0) * LDPC_FAST_STRIDE + (lane & 31)];      // the next layer's table (see `tev`)
                    C1_MARK(3);
                    // duplicate edges: ordered delta updates, level by level, by half A's lanes (conflict entry i is slot i: what it adds is od[i] = new - old)
                    if (ncf > 0) {
                        auto addr_of = [&](uint32_t e) __attribute__((always_inline)) { const uint32_t d = t4 - (e & 0x7FFu); return min(d, d + (uint32_t)C1_ROW) + ((e >> 11) & 0x3FFFFu); };
                        uint32_t prev_lvl = 0u;
#pragma unroll
                        for (int i = 0; i < LDPC_SPA_MAXC; i++) {
                            if (i >= ncf) break;
                            const uint32_t e = i == 0 ? ce0 : i == 1 ? ce1 : T[32 + i];
                            const uint32_t lvl = i == 0 ? 1u : i == 1 ? (cinfo >> 21) & 3u : T[48 + i] >> 8;
                            if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
                            if (HALF == 0 && act) { const uint32_t a = addr_of(e); const float Lv = lld(a); lst(a, Lv + od[i]); }
                        }
                    }
                    C1_MARK(4);
                    __syncthreads();
                    C1_MARK(5);
                    continue;
                }
                float v[NS];
                uint32_t w[NS];
                const float c1o = nx1, c2o = nx2;
                const uint32_t pko = __float_as_uint(nxk);
                float mn1 = INFINITY, mn2 = INFINITY, cst1 = 0.f, cst2 = 0.f;
                uint32_t sacc = 0u, tot = 0u, pkn = 0u;
                if (act) {
                    // ---- pass 1a: every posterior load of the half-check in flight before any use
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const uint32_t d = t4 - (E[j] & 0x7FFu);
                        const uint32_t base = (E[j] >> 11) & 0x3FFFFu;
                        w[j] = min(d, d + (uint32_t)C1_ROW) + base;
                        if (FWD && j == NS - 1 && r > 0) v[j] = pfw;         // p_{c-1}: handed over by layer r - 1
                        else v[j] = lld(w[j]);
                    }
                    const int rn = r + 1 < q ? r + 1 : 0;
                    if (it == 0 && r + 1 < q) { nx1 = 0.f; nx2 = 0.f; nxk = 0.f; }       // layer r + 1 has no messages yet in the first iteration
                    else {      // {c1, c2, pk A, pk B} of a check are 16 consecutive bytes
                        const uint32_t so = (uint32_t)(rn * LDPC_Z) * 16u;
                        if (HALF == 0) { const u32x3 sv = __builtin_amdgcn_raw_buffer_load_b96(rs, t4 * 4u, so, 0); nx1 = __uint_as_float(sv.x); nx2 = __uint_as_float(sv.y); nxk = __uint_as_float(sv.z); }
                        else { const u32x4 sv = __builtin_amdgcn_raw_buffer_load_b128(rs, t4 * 4u, so, 0); nx1 = __uint_as_float(sv.x); nx2 = __uint_as_float(sv.y); nxk = __uint_as_float(sv.w); }
                    }
                    C1_MARK(0);
                    // ---- pass 1b: v->c = posterior - old c->v ; running min1 / min2 / signs of this half
                    const uint32_t idxo = pko >> 27;
                    if (it == 0) {
#pragma unroll
                        for (int j = 0; j < NS; j++) {
                            float x = v[j];
                            if (HALF == 1 && j == NS - 1 && mask0) x = INFINITY;
                            v[j] = x;
                            const float a = fabsf(x);
                            mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                            mn1 = fminf(mn1, a);
                            sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(x), 31);
                        }
                    } else
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const float mag = (idxo == (uint32_t)j) ? c1o : c2o;
                        const float old = c1_and_or(pko << ((32u - NS) + j), SB, mag);          // sign bit of local slot j | magnitude (>= 0)
                        float x = v[j] - old;
                        if (HALF == 1 && j == NS - 1 && mask0) x = INFINITY;
                        v[j] = x;
                        const float a = fabsf(x);
                        mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                        mn1 = fminf(mn1, a);
                        sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(x), 31);      // shift the sign bit in
                    }
                    // this half's {min1 | parity of its signs, min2} for the other half (+inf | 0 stays +inf: the absent edge's sign bit is 0)
                    const uint32_t par = (uint32_t)(__popc(sacc) & 1);
                    f32x2 mine;
                    mine.x = __uint_as_float(__float_as_uint(mn1) | (par << 31)); mine.y = mn2;
                    *(lds_f32x2 *)(size_t)xmine = mine;
                    tot = par;
                }
                C1_MARK(1);
                __syncthreads();         // the halves' partial results are in place; every read of the layer precedes its writes
                C1_MARK(2);
                if (act) {
                    // ---- merge: the two smallest magnitudes of the whole check and the parity of all its signs
                    const f32x2 oth = *(lds_f32x2 *)(size_t)xother;
                    const float o1 = fabsf(oth.x), o2 = oth.y;
                    tot ^= __float_as_uint(oth.x) >> 31;
                    const float hi = fmaxf(mn1, o1);
                    mn1 = fminf(mn1, o1);
                    mn2 = fminf(fminf(mn2, o2), hi);
                    cst1 = mn2 * p.alpha; cst2 = mn1 * p.alpha;
                    pkn = sacc ^ (tot ? ((1u << NS) - 1u) : 0u);                              // sign(new_j) = tot ^ sign(x_j)
                }
                // the next layer's table: requested now (the slot entries of this layer are dead: pass 2 stores where pass 1 loaded) so that the scalar loads travel under
                // pass 2 instead of in front of the end-of-layer barrier
                __builtin_amdgcn_sched_barrier(0);
                {
                    const const_u32 Tn = tab + (r + 1 < q ? r + 1 : 0) * LDPC_FAST_STRIDE;
#pragma unroll
                    for (int j = 0; j < 32; j++) TE[j] = Tn[j];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (act) {
                    // ---- pass 2: new c->v ; posterior = v->c + new c->v.  Duplicate edges (not in `prim`) go to the junk row, the absent edge of lane 0 too
                    uint32_t idxn = 31u;                                                      // 31: the minimum is not among this half's slots
                    float m1s = __uint_as_float(__float_as_uint(cst1) | (tot << 31));         // output magnitudes carrying the total sign
                    float m2s = __uint_as_float(__float_as_uint(cst2) | (tot << 31));
                    asm volatile("" : "+v"(m1s), "+v"(m2s));
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const float x = v[j];
                        const bool ismin = fabsf(x) == mn1;
                        const float mag = ismin ? m1s : m2s;
                        const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & SB));
                        idxn = ismin ? (uint32_t)j : idxn;
                        asm("" : "+v"(idxn));
                        uint32_t a = w[j];
                        if (HALF == 1 && j == NS - 1 && mask0) a = ljunk + t4;
                        if (FWD && j == NS - 2 && r + 1 < q) pfw = x + nw;                    // p_c: kept for layer r + 1
                        else if (HALF == 0 && j < ldpc_w8_kd(DEG)) {      // only these slots can hold a duplicate edge (plan): its plain store is left out (a scalar branch)
                            if (((prim >> j) & 1u) != 0u) lst(a, x + nw);
                        } else lst(a, x + nw);
                    }
                    pkn |= idxn << 27;
                    {
                        const uint32_t so = (uint32_t)(r * LDPC_Z) * 16u;
                        if (HALF == 0) { u32x3 sv; sv.x = __float_as_uint(cst1); sv.y = __float_as_uint(cst2); sv.z = pkn; __builtin_amdgcn_raw_buffer_store_b96(sv, rs, wide_off(t4 * 4u, so), 0u, 0); }
                        else __builtin_amdgcn_raw_buffer_store_b32(pkn, rs, t4 * 4u + 12u, so, 0);
                    }
                    if (q == 1) { nx1 = cst1; nx2 = cst2; nxk = __uint_as_float(pkn); }
                }
                C1_MARK(3);
                // ---- duplicate edges of a bit-group inside this layer: ordered delta updates, level by level, by half A's lanes (conflict entry i is slot i)
                if (ncf > 0) {
                    auto addr_of = [&](uint32_t e) { const uint32_t d = t4 - (e & 0x7FFu); return min(d, d + (uint32_t)C1_ROW) + ((e >> 11) & 0x3FFFFu); };
                    auto delta_of = [&](uint32_t j) { return c1_unpack<NS>(cst1, cst2, pkn, j, SB) - c1_unpack<NS>(c1o, c2o, pko, j, SB); };
                    const uint32_t j0 = (cinfo >> 8) & 31u, j1 = (cinfo >> 16) & 31u, lvl1 = (cinfo >> 21) & 3u;
                    const bool two = ncf > 1 && lvl1 == 1u;         // entry 1 commutes with entry 0 (another bit-group)
                    __syncthreads();                                // the primary writes of the layer are in place
                    if (HALF == 0 && act) {
                        const uint32_t a0 = addr_of(ce0), a1 = addr_of(ce1);
                        float L0, L1 = 0.f;
                        L0 = lld(a0); if (two) L1 = lld(a1);
                        const float d0 = it == 0 ? c1_unpack<NS>(cst1, cst2, pkn, j0, SB) : delta_of(j0), d1 = it == 0 ? c1_unpack<NS>(cst1, cst2, pkn, j1, SB) : delta_of(j1);
                        lst(a0, L0 + d0); if (two) lst(a1, L1 + d1);
                    }
                    uint32_t prev_lvl = 1u;
                    for (int i = two ? 2 : 1; i < ncf; i++) {
                        const uint32_t e = T[32 + i], meta = T[48 + i];
                        const uint32_t j = meta & 31u, lvl = meta >> 8;
                        if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
                        if (HALF == 0 && act) {
                            const uint32_t a = addr_of(e);
                            const float Lv = lld(a);
                            lst(a, Lv + (it == 0 ? c1_unpack<NS>(cst1, cst2, pkn, j, SB) : delta_of(j)));
                        }
                    }
                }
                C1_MARK(4);
                __syncthreads();
                C1_MARK(5);
            }
            it++;
            if (p.early_stop || it == p.n_ite) {
                // ---- syndrome of the hard decisions, layer by layer (half A's lanes read all DEG slots of their check; the row-keeping waves go on moving rows,
                //      so every sweep layer ends at a barrier; under the stopping rule layer 0 is voted on before anything moves): k_ldpc_wg8.hip, parked modes
                ok = true;
                int bad = 0;
                auto vote = [&](int b) -> bool {
                    const bool any = __ballot(b != 0) != 0ull;
                    lds_int *const wv = s_misc + 20;
                    const int k = nvote % 3;
                    nvote++;
                    if (lane == 0) { if (any) wv[k] = 1; if (vote_writer) wv[(k + 1) % 3] = 0; }
                    __syncthreads();
                    return wv[k] != 0;
                };
                for (int r = 0; r < q; r++) {
                    if (HALF == 0 && act) {
                        const const_u32 T = tab + r * LDPC_FAST_STRIDE;
                        float Lv[DEG];
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            const uint32_t e = T[j];
                            const uint32_t d = t4 - (e & 0x7FFu);
                            Lv[j] = lld(min(d, d + (uint32_t)C1_ROW) + ((e >> 11) & 0x3FFFFu));
                        }
                        if (r == 0 && t == 0) Lv[DEG - 1] = 0.f;                               // absent edge
                        uint32_t x = 0u;
#pragma unroll
                        for (int j = 0; j < DEG; j++) x ^= __float_as_uint(Lv[j]);
                        bad |= (int)(x >> 31);
                    }
                    if (p.early_stop) {
                        if (vote(bad)) { ok = false; if (r > 0) __syncthreads(); break; }
                        if (r == 0) __syncthreads();
                    } else if (r == q - 1) { if (vote(bad)) ok = false; }
                    else __syncthreads();
                }
                C1_MARK(6);
                if (ok) break;
            }
        }

        // ---- outputs: hard decisions of the info bits (rows in LDS position order, the halves take alternate batches; then the rows the row-keeping waves hand back)
        if (first_wave && lane == 0) {
            if (p.cwd) p.cwd[f] = ok ? 1 : 0;
            if (p.ites) p.ites[f] = it;
        }
        const int n_words = (p.K + 31) / 32;
        const const_u32 prbs_c = (const_u32)p.info_prbs;
        auto emit = [&](int g, float Lv) {
            if (p.info_out && g < p.n_info) {
                // fused chain: descrambled info bits as int32; the 64 PRBS bits of this wave's stretch of the row come in as one wave-uniform 64-bit scalar word
                const int kb = g * LDPC_Z + cb * 64;
                const const_u32 pq = prbs_c + 2 * (g * 6 + cb);
                const unsigned long long m64 = (unsigned long long)pq[0] | ((unsigned long long)pq[1] << 32);
                const int k = kb + lane;
                if (act && k < p.K_info)
                    __builtin_nontemporal_store((int32_t)((Lv < 0.f ? 1u : 0u) ^ (uint32_t)((m64 >> lane) & 1ull)), &p.info_out[(size_t)f * p.K_info + k]);
            }
            if (act) {
                if (g < p.n_info) {
                    if (p.bits) __builtin_nontemporal_store((int32_t)(Lv < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + g * LDPC_Z + t]);
                    if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + t] = Lv;
                } else if (p.post) p.post[(size_t)f * p.N + p.K + q * t + (g - p.n_info)] = Lv;
            }
            if (p.packed && g < p.n_info) {
                const unsigned long long m = __ballot(act && Lv < 0.f);
                const int o8 = g * (LDPC_Z / 8) + cb * 8, nb = cb * 64 + 64 <= LDPC_Z ? 8 : (LDPC_Z - cb * 64) / 8;
                if (lane < nb) reinterpret_cast<uint8_t *>(p.packed + (size_t)f * n_words)[o8 + lane] = (uint8_t)(m >> (8 * lane));
            }
        };
        const int nl_out = nl;      // (emit() drops what the caller does not want: the parity groups unless the posteriors are asked for)
        for (int l0 = HALF * C1_IO; l0 < nl_out; l0 += 2 * C1_IO) {
            float v[C1_IO];
#pragma unroll
            for (int k = 0; k < C1_IO; k++) v[k] = act ? lld((uint32_t)(l0 + k < nl_out ? l0 + k : nl_out - 1) * C1_ROW + t4) : 0.f;
#pragma unroll
            for (int k = 0; k < C1_IO; k++) if (l0 + k < nl_out) emit((int)rows[l0 + k], v[k]);
        }
        {
            constexpr int NRT = 2 * ldpc_cu1_nrg();
            __syncthreads();
            __syncthreads();
            const const_u32 srow = rows + nl + p.w8.ng + q;
            for (int l0 = HALF * C1_IO; l0 < NRT; l0 += 2 * C1_IO) {
                float v[C1_IO];
#pragma unroll
                for (int k = 0; k < C1_IO; k++) v[k] = act ? lld((uint32_t)(l0 + k < NRT ? l0 + k : NRT - 1) * C1_ROW + t4) : 0.f;
#pragma unroll
                for (int k = 0; k < C1_IO; k++) if (l0 + k < NRT && srow[l0 + k] != 0xFFFFFFFFu) emit((int)srow[l0 + k], v[k]);
            }
        }
        if (first_wave && lane == 0) s_misc[19] = p.cu_ctr ? (int)(atomicAdd(&p.cu_ctr[LDPC_FRAME_CTR], 1u) + gridDim.x) : qp + (int)gridDim.x;
        __syncthreads();     // the posterior image is reused by the next frame of this workgroup
        qp = s_misc[19];
        C1_MARK(7);
#ifdef LDPC_PHASE_PROF
        prof[9]++;
#endif
    }
}

template <int DEG, int SPA>
__global__ void __launch_bounds__(C1_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
ldpc_cu1_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    if ((uint32_t)(size_t)(lds_float *)smem != 0u) __builtin_trap();          // LDS is addressed by plain byte offsets (k_ldpc_wg8.hip)
    lds_int *const s_misc = (lds_int *)c1_lds((uint32_t)p.w8.lds_bytes - 128u);      // [0..15] SIMD of wave w, [19] next frame, [20..22] vote words
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);            // HW_REG_HW_ID
    const int simd = (int)((hw >> 4) & 3u);
    if (lane == 0) s_misc[wave] = simd;
    if (threadIdx.x == 0) { s_misc[20] = 0; s_misc[21] = 0; s_misc[22] = 0; }
    __syncthreads();
    // roles: every SIMD carries three working waves and one row-keeping wave (the dispatcher deals a workgroup's waves round-robin over the four SIMDs; should
    // it ever not, the roles follow the wave index: a performance matter only).  Working role w = 0 .. 11: half w & 1, check block w >> 1; keeper k = 0 .. 3: group k >> 1.
    int role, keeper = -1;
    {
        int k = 0, c[4] = {0, 0, 0, 0};
        for (int w = 0; w < C1_WAVES; w++) {
            const int s = s_misc[w];
            c[0] += s == 0; c[1] += s == 1; c[2] += s == 2; c[3] += s == 3;
            if (w < wave && s == simd) k++;
        }
        const bool balanced = c[0] == 4 && c[1] == 4 && c[2] == 4 && c[3] == 4;
        if (!balanced) { role = wave < 12 ? wave : -1; keeper = wave < 12 ? -1 : wave - 12; }
        else if (k < 3) role = simd * 3 + k;
        else { role = -1; keeper = simd; }
        role = __builtin_amdgcn_readfirstlane(role); keeper = __builtin_amdgcn_readfirstlane(keeper);
    }
#ifdef LDPC_PHASE_PROF
    uint32_t prof[12];
    for (int i = 0; i < 12; i++) prof[i] = 0u;
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long pt_ = prof_t0;
#endif
    if (role < 0) {
#ifdef LDPC_PHASE_PROF
        cu1_keeper<ldpc_cu1_nrg(), SPA>(p, s_misc, keeper >> 1, keeper & 1, lane, keeper == 0, prof, pt_);
        if (lane == 0 && p.cu_ctr) {
            prof[10] = 1u; prof[11] = (uint32_t)(__builtin_amdgcn_s_memtime() - prof_t0);
            for (int i = 0; i < 12; i++) p.cu_ctr[LDPC_CU_CTR_WORDS + ((int)blockIdx.x * C1_WAVES + wave) * 12 + i] = prof[i];
        }
#else
        cu1_keeper<ldpc_cu1_nrg(), SPA>(p, s_misc, keeper >> 1, keeper & 1, lane, keeper == 0);
#endif
        return;
    }
    // (the vote words are zeroed by keeper 0's lane 0; the frame counter and the flags are written by working role 0's lane 0)
#ifdef LDPC_PHASE_PROF
    if (role & 1) cu1_work<DEG, 1, SPA, SPA ? C1_HA_SPA : C1_HA>(p, s_misc, role >> 1, lane, wave, false, false, prof, pt_);
    else cu1_work<DEG, 0, SPA, SPA ? C1_HA_SPA : C1_HA>(p, s_misc, role >> 1, lane, wave, false, role == 0, prof, pt_);
    if (lane == 0 && p.cu_ctr) {
        prof[10] = 0x100u + (uint32_t)role; prof[11] = (uint32_t)(__builtin_amdgcn_s_memtime() - prof_t0);
        for (int i = 0; i < 12; i++) p.cu_ctr[LDPC_CU_CTR_WORDS + ((int)blockIdx.x * C1_WAVES + wave) * 12 + i] = prof[i];
    }
#else
    if (role & 1) cu1_work<DEG, 1, SPA, SPA ? C1_HA_SPA : C1_HA>(p, s_misc, role >> 1, lane, wave, false, false);
    else cu1_work<DEG, 0, SPA, SPA ? C1_HA_SPA : C1_HA>(p, s_misc, role >> 1, lane, wave, false, role == 0);
#endif
}

hipError_t ldpc_cu1_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    p.cu_ctr = pl.d_cu_ctr;
    p.w8.tab = pl.d_w8_tab; p.w8.rows = pl.d_w8_rows; p.w8.atab = nullptr;
    p.w8.st_base = pl.w8_st_base; p.w8.lds_junk = pl.w8_lds_junk; p.w8.lds_bytes = pl.w8_lds_bytes; p.w8.pad = 0;
    p.w8.nl_info = pl.w8_nl_info; p.w8.nl = pl.w8_nl; p.w8.ng_info = pl.w8_ng_info; p.w8.ng = pl.w8_ng;
    p.spa_cap = pl.spa_rule == 3 ? LDPC_SPA_CAP : INFINITY;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    p.gwork_words = pl.w8_gwork_words;
    if (pl.fast_deg != 27 || pl.cu1_pairs != 2 * ldpc_cu1_nrg()) return hipErrorInvalidValue;
    auto kern = pl.spa_rule == 3 ? ldpc_cu1_kernel<27, 3> : pl.spa ? ldpc_cu1_kernel<27, 1> : ldpc_cu1_kernel<27, 0>;
    static size_t configured_dev[3][64] = {{0}, {0}, {0}};
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t &configured = configured_dev[pl.spa_rule == 3 ? 2 : pl.spa ? 1 : 0][dev & 63];
    const size_t lds = (size_t)pl.w8_lds_bytes;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    const int grid = p.n_frames < pl.grid_max ? p.n_frames : pl.grid_max;
    if (p.cu_ctr) { hipError_t e = hipMemsetAsync(p.cu_ctr, 0, LDPC_CU_CTR_WORDS * sizeof(uint32_t), s); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C1_THREADS), lds, s, p);
#ifdef LDPC_PHASE_PROF
    {   // development aid: average ticks per phase over the working waves
        (void)hipStreamSynchronize(s);
        std::vector<uint32_t> hbuf((size_t)grid * C1_WAVES * 12);
        (void)hipMemcpy(hbuf.data(), p.cu_ctr + LDPC_CU_CTR_WORDS, hbuf.size() * 4, hipMemcpyDeviceToHost);
        double acc[12] = {0}, kacc[12] = {0}, racc[12][12] = {{0}}; int nw = 0, nk = 0, rn[12] = {0}; double fr = 0;
        for (int w = 0; w < grid * C1_WAVES; w++) {
            if (!hbuf[(size_t)w * 12 + 11]) continue;
            if (hbuf[(size_t)w * 12 + 10] == 1u) { nk++; for (int i = 0; i < 12; i++) kacc[i] += hbuf[(size_t)w * 12 + i]; continue; }
            nw++; for (int i = 0; i < 12; i++) acc[i] += hbuf[(size_t)w * 12 + i]; fr += hbuf[(size_t)w * 12 + 9];
            const int role = (int)(hbuf[(size_t)w * 12 + 10] & 0xFFu);
            if (role < 12) { for (int i = 0; i < 12; i++) racc[role][i] += hbuf[(size_t)w * 12 + i]; rn[role]++; }
        }
        for (int role = 0; role < 12; role++) if (rn[role]) {
            const double dn = racc[role][9] * pl.q * p.n_ite;
            fprintf(stderr, "  role %2d (half %c, checks %3d..): per layer 1a %5.0f 1b %5.0f merge-wait %5.0f pass2 %5.0f replay %5.0f end-wait %5.0f\n", role, role & 1 ? 'B' : 'A', (role >> 1) * 64,
                    racc[role][0] / dn, racc[role][1] / dn, racc[role][2] / dn, racc[role][3] / dn, racc[role][4] / dn, racc[role][5] / dn);
        }
        if (nk) fprintf(stderr, "[ldpc cu1 phase prof] %d row-keeping waves: swaps + inner barriers %.0f, waiting at the end barrier %.0f ticks per layer (of %.0f per wave)\n", nk,
                        kacc[0] / kacc[9] / (pl.q * p.n_ite), kacc[1] / kacc[9] / (pl.q * p.n_ite), kacc[11] / nk);
        static const char *nm[12] = {"1a issue", "1b", "merge barrier", "pass 2", "replay", "end barrier", "syndrome", "output", "input", "-frames", "-", "TOTAL"};
        fprintf(stderr, "[ldpc cu1 phase prof] %d working waves, %.2f frames per wave, %.0f ticks per frame and layer (q %d, %d iterations)\n", nw, fr / (nw ? nw : 1),
                (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5]) / (fr > 0 ? fr : 1) / (pl.q * p.n_ite), pl.q, p.n_ite);
        for (int i = 0; i < 12; i++) if (nm[i][0] != '-') fprintf(stderr, "  %-14s %12.0f ticks/wave  %5.1f %%\n", nm[i], acc[i] / (nw ? nw : 1), 100.0 * acc[i] / (acc[11] > 0 ? acc[11] : 1));
    }
#endif
    return hipGetLastError();
}

}  // namespace dvbs2
