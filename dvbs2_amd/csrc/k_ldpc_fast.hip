// a1 fast path -- layered NMS for the REGULAR DVB-S2 codes (every check has the same degree:
// 27 for rate 8/9, 11 for rate 3/5), same schedule and arithmetic as k_ldpc.hip, bit-exact
// with it and with the oracle, but built around what bounds this kernel on gfx950:
//
//   * the CU's single scalar unit.  The generic kernel spends ~70 SALU instructions per edge
//     (table unpacking, LDS-vs-global branches, flag tests) and is SALU-issue-bound.  Here a
//     layer costs 2 SALU per edge: the layer table is one dword per slot (byte shift | byte
//     offset of the bit-group), preloaded with wide s_loads; there is no per-edge branch --
//     the few same-layer duplicate edges are redirected with a select (out-of-range buffer
//     offset / dummy LDS row) in pass 2 and replayed from a short per-layer list afterwards;
//   * address VALU.  Per-lane byte offsets w_j = ((t - t0_j) mod 360) * 4 are computed once
//     per layer and kept in VGPRs for the store pass; global posteriors go through ONE raw
//     buffer descriptor (32-bit voffset + SGPR soffset, no 64-bit address math);
//   * latency.  All 27 posterior loads of a check are issued before the first is used; the
//     next layer's packed c->v state (private to the lane) is prefetched under the compute;
//     and (NF = 2) every lane works on the same check of TWO frames at once: two independent
//     dependency chains per lane, table unpacking and address arithmetic paid once for both.
//
// Posterior image: bit-group g at word 360 g (info groups, then parity groups regrouped
// [r][t]), either entirely in LDS (N = 16200: 64.8 KB, two workgroups per CU) or entirely
// in the workgroup's global workspace slot (N = 64800), which is reused frame after frame
// and therefore stays L2 / Infinity-Cache resident.  Packed c->v state always lives in the
// workspace: 12 B per check per layer, coalesced, prefetched.
#include "dvbs2hip_internal.h"
#include <cstdlib>

namespace dvbs2 {

typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(4))) uint32_t *const_u32;
typedef const __attribute__((address_space(4))) unsigned long long *const_u64;
constexpr uint32_t FE_LDS = 1u << 29;       // fast-table entry: the bit-group lives in LDS (hybrid modes)
// MODE 3 = STATIC hybrid: the host picks the LDS-resident bit-groups so that EVERY layer has exactly
// NL_STATIC of its 27 slots in LDS and sorts them first: slot j < NL_STATIC is an LDS access, the
// others are buffer accesses -- decided at compile time, no per-slot branch, select or dual issue.
constexpr int NL_STATIC = 9;
__host__ __device__ constexpr bool slot_in_lds(int mode, int j) { return mode == 0 || (mode == 3 && j < NL_STATIC); }

constexpr uint32_t OOB = 0x7FFFF000u;       // beyond every workspace: loads return 0, stores are dropped
constexpr int ROW_BYTES = LDPC_Z * 4;
#ifndef ST_AUX
#define ST_AUX 0        // cache policy of the packed-state traffic (2 = nt)
#endif

// Packed per-check state of the fast path: bits 31..27 = slot of the minimum, bit (DEG-1-j) = sign of
// the message on slot j (the order v_alignbit shifts them in).  Decompresses to the exact fp32 message.
template <int DEG>
__device__ __forceinline__ float c2v_unpack_dyn(float c1, float c2, uint32_t pk, uint32_t j)
{
    const float mag = ((pk >> 27) == j) ? c1 : c2;
    return __uint_as_float(__float_as_uint(mag) | ((pk << ((32u - DEG) + j)) & 0x80000000u));
}

template <int MODE>
struct FastCtx {
    __amdgpu_buffer_rsrc_t rs;   // the workgroup's whole workspace slot
    lds_float *lpost;            // MODE 0: posterior image in LDS
    const_u32 tab;               // layer table
    uint32_t t4;                 // 4 * lane
    uint32_t c2v_base;           // byte offset of the packed state inside one frame's workspace
    uint32_t redirect;           // where dropped stores go (OOB voffset / LDS dummy row)
    uint32_t zero_row, junk_row; // MODE 2: byte offsets of the always-zero and the write-only LDS rows
    int M, q;
    float alpha;

    __device__ __forceinline__ float post_ld(uint32_t off, uint32_t soff) const
    {
        if (MODE == 0) return lpost[(off + soff) >> 2];
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, soff, 0));
    }
    __device__ __forceinline__ void post_st(uint32_t off, uint32_t soff, float v) const
    {
        if (MODE == 0) lpost[(off + soff) >> 2] = v;
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, off, soff, 0);
    }
    // slot-indexed access: after unrolling, j is a constant and MODE 3's `j < NL_STATIC` folds away
    __device__ __forceinline__ float slot_ld(int j, uint32_t w, uint32_t e, uint32_t fo) const
    {
        if (MODE == 3) {
            const uint32_t base = (e >> 11) & 0x3FFFFu;
            if (j < NL_STATIC) return lpost[(w + base) >> 2];
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, w, base + fo, 0));
        }
        return post_ld(w, (MODE == 0 ? 0u : (e >> 11)) + fo);
    }
    __device__ __forceinline__ void slot_st(int j, uint32_t w, bool keep, uint32_t e, uint32_t fo, float v) const
    {
        if (MODE == 3) {
            const uint32_t base = (e >> 11) & 0x3FFFFu;
            if (j < NL_STATIC) lpost[(keep ? w + base : junk_row + t4) >> 2] = v;
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, keep ? w : OOB, base + fo, 0);
        } else post_st(keep ? w : redirect, (MODE == 0 ? 0u : (e >> 11)) + fo, v);
    }
    __device__ __forceinline__ float st_ld(uint32_t fo, int arr, int r) const
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t4, fo + c2v_base + (uint32_t)(arr * M + r * LDPC_Z) * 4u, ST_AUX));
    }
    __device__ __forceinline__ void st_st(uint32_t fo, int arr, int r, float v) const
    {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, t4, fo + c2v_base + (uint32_t)(arr * M + r * LDPC_Z) * 4u, ST_AUX);
    }
};

// one decoding iteration (all q layers) for NA frames whose workspaces start at byte offsets fo[]
template <int DEG, int MODE, int NA>
__device__ __forceinline__ void fast_iteration(const FastCtx<MODE> &c, const uint32_t (&fo)[NA], float (&nx)[NA][3], bool act, int t)
{
    const int q = c.q;
    for (int r = 0; r < q; r++) {
        const const_u32 T = c.tab + r * LDPC_FAST_STRIDE;
        uint32_t E[DEG];
#pragma unroll
        for (int j = 0; j < DEG; j++) E[j] = T[j];
        const uint32_t prim = T[27];
        const int ncf = (int)T[28];
        const bool mask0 = (r == 0) && (t == 0);        // p_{c-1} of check 0 does not exist
        float v[NA][DEG];
        uint32_t w[DEG];
        float c1o[NA], c2o[NA], cst1[NA], cst2[NA];
        uint32_t pko[NA], pkn[NA];
        if (act) {
            // ---- pass 1a: every posterior load of the check(s) in flight before any use
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const uint32_t d = c.t4 - (E[j] & 0x7FFu);
                if (MODE == 2) {
                    // hybrid image, no branch: issue BOTH an LDS read and a buffer load; the one
                    // that does not apply hits the all-zero LDS row / an out-of-range offset
                    const bool il = (E[j] & FE_LDS) != 0u;
                    const uint32_t base = (E[j] >> 11) & 0x3FFFFu;
                    w[j] = min(d, d + (uint32_t)ROW_BYTES);
                    const float a = c.lpost[(w[j] + (il ? base : c.zero_row)) >> 2];
                    const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, w[j] + (il ? OOB : 0u), base + fo[0], 0));
                    v[0][j] = il ? a : b;
                } else {
                    w[j] = min(d, d + (uint32_t)ROW_BYTES) + (MODE == 0 ? (E[j] >> 11) : 0u);   // LDS: full address
#pragma unroll
                    for (int k = 0; k < NA; k++) v[k][j] = c.slot_ld(j, w[j], E[j], fo[k]);
                }
            }
            const int rn = r + 1 < q ? r + 1 : 0;
#pragma unroll
            for (int k = 0; k < NA; k++) {
                c1o[k] = nx[k][0]; c2o[k] = nx[k][1]; pko[k] = __float_as_uint(nx[k][2]);
                // prefetch the next layer's packed state (private to this lane)
                nx[k][0] = c.st_ld(fo[k], 0, rn); nx[k][1] = c.st_ld(fo[k], 1, rn); nx[k][2] = c.st_ld(fo[k], 2, rn);
            }
        }
        // ---- pass 1b: v->c = posterior - old c->v ; running min1 / min2 / sign
        float mn1[NA], mn2[NA];
        uint32_t sacc[NA], tot[NA];
#pragma unroll
        for (int k = 0; k < NA; k++) { mn1[k] = INFINITY; mn2[k] = INFINITY; sacc[k] = 0u; tot[k] = 0u; cst1[k] = 0.f; cst2[k] = 0.f; pkn[k] = 0u; }
        if (act) {
#pragma unroll
            for (int j = 0; j < DEG; j++) {
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    float x = v[k][j] - c2v_unpack_dyn<DEG>(c1o[k], c2o[k], pko[k], (uint32_t)j);
                    if (j == DEG - 1 && mask0) x = INFINITY;
                    v[k][j] = x;
                    const float a = fabsf(x);
                    mn2[k] = __builtin_amdgcn_fmed3f(mn1[k], mn2[k], a);
                    mn1[k] = fminf(mn1[k], a);
                    sacc[k] = __builtin_amdgcn_alignbit(sacc[k], __float_as_uint(x), 31);      // shift the sign bit in
                }
            }
#pragma unroll
            for (int k = 0; k < NA; k++) {
                cst1[k] = mn2[k] * c.alpha; cst2[k] = mn1[k] * c.alpha;
                // all new signs at once: sign(new_j) = (parity of all signs) ^ sign(x_j)
                tot[k] = (uint32_t)(__popc(sacc[k]) & 1);
                pkn[k] = sacc[k] ^ (tot[k] ? ((1u << DEG) - 1u) : 0u);
            }
        }
        if (ncf > 0) __syncthreads();         // every read of the layer precedes its writes
        if (act) {
            // ---- pass 2: new c->v ; posterior = v->c + new c->v.  Duplicate edges (not in
            //      `prim`) and the absent edge are redirected, not branched around.
            uint32_t idxn[NA];
            float m1s[NA], m2s[NA];       // output magnitudes carrying the total sign
#pragma unroll
            for (int k = 0; k < NA; k++) {
                idxn[k] = 0u;
                m1s[k] = __uint_as_float(__float_as_uint(cst1[k]) | (tot[k] << 31));
                m2s[k] = __uint_as_float(__float_as_uint(cst2[k]) | (tot[k] << 31));
            }
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const bool keep = ((prim >> j) & 1u) != 0u && !(j == DEG - 1 && mask0);     // else: store dropped / redirected
                const uint32_t off = keep ? w[j] : c.redirect;
                uint32_t la = 0u, go = 0u, hbase = 0u;
                if (MODE == 2) {
                    const bool il = (E[j] & FE_LDS) != 0u, pr = ((prim >> j) & 1u) != 0u;
                    hbase = (E[j] >> 11) & 0x3FFFFu;
                    la = w[j] + ((il && pr) ? hbase : c.junk_row);
                    go = w[j] + ((!il && pr) ? 0u : OOB);
                    if (j == DEG - 1 && mask0) { la = c.junk_row + c.t4; go = OOB; }
                }
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    const float x = v[k][j];
                    const bool ismin = fabsf(x) == mn1[k];
                    const float mag = ismin ? m1s[k] : m2s[k];
                    const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & 0x80000000u));
                    idxn[k] = ismin ? (uint32_t)j : idxn[k];
                    if (MODE == 2) {
                        c.lpost[la >> 2] = x + nw;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x + nw), c.rs, go, hbase + fo[0], 0);
                    } else if (MODE == 3) c.slot_st(j, w[j], keep, E[j], fo[k], x + nw);
                    else c.post_st(off, (MODE == 0 ? 0u : (E[j] >> 11)) + fo[k], x + nw);
                }
            }
#pragma unroll
            for (int k = 0; k < NA; k++) {
                pkn[k] |= idxn[k] << 27;
                c.st_st(fo[k], 0, r, cst1[k]); c.st_st(fo[k], 1, r, cst2[k]); c.st_st(fo[k], 2, r, __uint_as_float(pkn[k]));
                if (q == 1) { nx[k][0] = cst1[k]; nx[k][1] = cst2[k]; nx[k][2] = __uint_as_float(pkn[k]); }
            }
        }
        // ---- duplicate edges of a bit-group inside this layer: ordered delta updates
        uint32_t prev_lvl = 0u;
        for (int i = 0; i < ncf; i++) {
            const uint32_t e = T[32 + i], meta = T[48 + i];
            const uint32_t j = meta & 31u, lvl = meta >> 8;
            if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
            if (act) {
                const uint32_t d = c.t4 - (e & 0x7FFu);
                const uint32_t off = min(d, d + (uint32_t)ROW_BYTES);
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    const float nw = c2v_unpack_dyn<DEG>(cst1[k], cst2[k], pkn[k], j);
                    const float od = c2v_unpack_dyn<DEG>(c1o[k], c2o[k], pko[k], j);
                    if (MODE >= 2) {
                        const uint32_t hb = (e >> 11) & 0x3FFFFu;
                        if (e & FE_LDS) { const float L = c.lpost[(off + hb) >> 2]; c.lpost[(off + hb) >> 2] = L + (nw - od); }
                        else {
                            const float L = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, off, hb + fo[0], 0));
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, L + (nw - od)), c.rs, off, hb + fo[0], 0);
                        }
                    } else {
                        const float L = c.post_ld(off, (e >> 11) + fo[k]);
                        c.post_st(off, (e >> 11) + fo[k], L + (nw - od));
                    }
                }
            }
        }
        __syncthreads();
    }
}


// ---- software-pipelined iteration (one frame per lane).  EXPERIMENTAL, off by default: bit-exact, but on
// MI355X it lost to the plain iteration (N=16200 3/5: 4.8 vs 3.4 ms; at DEG 27 it needs > 256 VGPRs and
// spills: 49.8 vs 11.5 ms), see DESIGN.md section 6.  Host-side, the slots of every layer are
// sorted "early first": a slot is EARLY when its bit-group is not touched by the previous layer, so
// its posterior can be loaded -- and folded into min1/min2/sign, which do not depend on the order --
// before the previous layer's stores have drained.  Per layer r:
//     early loads of layer r+1  ->  pass 2 of layer r (stores)  ->  pass 1b of r+1 on the early slots
//     ->  barrier  ->  late loads of r+1  ->  pass 1b on the late slots
// so the store drain, the barrier and most of the load latency sit under arithmetic.  Partial slot
// ranges run as suffixes of unrolled sequences entered through one jump (no per-slot branch).
template <int DEG>
struct PipeBuf {
    float v[DEG];        // posterior, then v->c
    uint32_t w[DEG];     // byte offset of the posterior (bit-group base folded in)
    float c1o, c2o, mn1, mn2;
    uint32_t pko, sgn;   // old packed state; collected sign bits (bit DEG-1-j = sign of slot j)
};

template <int DEG, int MODE>
__device__ __forceinline__ void pipe_addr(const FastCtx<MODE> &c, const_u32 T, PipeBuf<DEG> &X)
{
#pragma unroll
    for (int j = 0; j < DEG; j++) {
        const uint32_t e = T[j];
        const uint32_t d = c.t4 - (e & 0x7FFu);
        X.w[j] = min(d, d + (uint32_t)ROW_BYTES) + (e >> 11);
    }
}

#define PIPE_LD(j) X.v[j] = c.post_ld(X.w[j], fo)
#define PIPE_1B(j)                                                                                     \
    do {                                                                                               \
        float x_ = X.v[j] - c2v_unpack_dyn<DEG>(X.c1o, X.c2o, X.pko, (uint32_t)(j));                   \
        if ((j) == DEG - 1 && m0) x_ = INFINITY;                                                       \
        X.v[j] = x_;                                                                                   \
        const float a_ = fabsf(x_);                                                                    \
        X.mn2 = __builtin_amdgcn_fmed3f(X.mn1, X.mn2, a_);                                             \
        X.mn1 = fminf(X.mn1, a_);                                                                      \
        X.sgn |= (__float_as_uint(x_) >> 31) << (DEG - 1 - (j));                                       \
    } while (0)

template <int DEG, int MODE>
__device__ __forceinline__ void pipe_loads_from(const FastCtx<MODE> &c, uint32_t fo, PipeBuf<DEG> &X, int first)
{
    switch (first) {
            case 0: if (0 < DEG) { PIPE_LD(0); } [[fallthrough]];
            case 1: if (1 < DEG) { PIPE_LD(1); } [[fallthrough]];
            case 2: if (2 < DEG) { PIPE_LD(2); } [[fallthrough]];
            case 3: if (3 < DEG) { PIPE_LD(3); } [[fallthrough]];
            case 4: if (4 < DEG) { PIPE_LD(4); } [[fallthrough]];
            case 5: if (5 < DEG) { PIPE_LD(5); } [[fallthrough]];
            case 6: if (6 < DEG) { PIPE_LD(6); } [[fallthrough]];
            case 7: if (7 < DEG) { PIPE_LD(7); } [[fallthrough]];
            case 8: if (8 < DEG) { PIPE_LD(8); } [[fallthrough]];
            case 9: if (9 < DEG) { PIPE_LD(9); } [[fallthrough]];
            case 10: if (10 < DEG) { PIPE_LD(10); } [[fallthrough]];
            case 11: if (11 < DEG) { PIPE_LD(11); } [[fallthrough]];
            case 12: if (12 < DEG) { PIPE_LD(12); } [[fallthrough]];
            case 13: if (13 < DEG) { PIPE_LD(13); } [[fallthrough]];
            case 14: if (14 < DEG) { PIPE_LD(14); } [[fallthrough]];
            case 15: if (15 < DEG) { PIPE_LD(15); } [[fallthrough]];
            case 16: if (16 < DEG) { PIPE_LD(16); } [[fallthrough]];
            case 17: if (17 < DEG) { PIPE_LD(17); } [[fallthrough]];
            case 18: if (18 < DEG) { PIPE_LD(18); } [[fallthrough]];
            case 19: if (19 < DEG) { PIPE_LD(19); } [[fallthrough]];
            case 20: if (20 < DEG) { PIPE_LD(20); } [[fallthrough]];
            case 21: if (21 < DEG) { PIPE_LD(21); } [[fallthrough]];
            case 22: if (22 < DEG) { PIPE_LD(22); } [[fallthrough]];
            case 23: if (23 < DEG) { PIPE_LD(23); } [[fallthrough]];
            case 24: if (24 < DEG) { PIPE_LD(24); } [[fallthrough]];
            case 25: if (25 < DEG) { PIPE_LD(25); } [[fallthrough]];
            case 26: if (26 < DEG) { PIPE_LD(26); } [[fallthrough]];
            default: break;
    }
}
template <int DEG, int MODE>
__device__ __forceinline__ void pipe_loads_below(const FastCtx<MODE> &c, uint32_t fo, PipeBuf<DEG> &X, int count)
{
    switch (count) {
            case 27: if (26 < DEG) { PIPE_LD(26); } [[fallthrough]];
            case 26: if (25 < DEG) { PIPE_LD(25); } [[fallthrough]];
            case 25: if (24 < DEG) { PIPE_LD(24); } [[fallthrough]];
            case 24: if (23 < DEG) { PIPE_LD(23); } [[fallthrough]];
            case 23: if (22 < DEG) { PIPE_LD(22); } [[fallthrough]];
            case 22: if (21 < DEG) { PIPE_LD(21); } [[fallthrough]];
            case 21: if (20 < DEG) { PIPE_LD(20); } [[fallthrough]];
            case 20: if (19 < DEG) { PIPE_LD(19); } [[fallthrough]];
            case 19: if (18 < DEG) { PIPE_LD(18); } [[fallthrough]];
            case 18: if (17 < DEG) { PIPE_LD(17); } [[fallthrough]];
            case 17: if (16 < DEG) { PIPE_LD(16); } [[fallthrough]];
            case 16: if (15 < DEG) { PIPE_LD(15); } [[fallthrough]];
            case 15: if (14 < DEG) { PIPE_LD(14); } [[fallthrough]];
            case 14: if (13 < DEG) { PIPE_LD(13); } [[fallthrough]];
            case 13: if (12 < DEG) { PIPE_LD(12); } [[fallthrough]];
            case 12: if (11 < DEG) { PIPE_LD(11); } [[fallthrough]];
            case 11: if (10 < DEG) { PIPE_LD(10); } [[fallthrough]];
            case 10: if (9 < DEG) { PIPE_LD(9); } [[fallthrough]];
            case 9: if (8 < DEG) { PIPE_LD(8); } [[fallthrough]];
            case 8: if (7 < DEG) { PIPE_LD(7); } [[fallthrough]];
            case 7: if (6 < DEG) { PIPE_LD(6); } [[fallthrough]];
            case 6: if (5 < DEG) { PIPE_LD(5); } [[fallthrough]];
            case 5: if (4 < DEG) { PIPE_LD(4); } [[fallthrough]];
            case 4: if (3 < DEG) { PIPE_LD(3); } [[fallthrough]];
            case 3: if (2 < DEG) { PIPE_LD(2); } [[fallthrough]];
            case 2: if (1 < DEG) { PIPE_LD(1); } [[fallthrough]];
            case 1: if (0 < DEG) { PIPE_LD(0); } [[fallthrough]];
            default: break;
    }
}
template <int DEG>
__device__ __forceinline__ void pipe_1b_from(PipeBuf<DEG> &X, bool m0, int first)
{
    switch (first) {
            case 0: if (0 < DEG) { PIPE_1B(0); } [[fallthrough]];
            case 1: if (1 < DEG) { PIPE_1B(1); } [[fallthrough]];
            case 2: if (2 < DEG) { PIPE_1B(2); } [[fallthrough]];
            case 3: if (3 < DEG) { PIPE_1B(3); } [[fallthrough]];
            case 4: if (4 < DEG) { PIPE_1B(4); } [[fallthrough]];
            case 5: if (5 < DEG) { PIPE_1B(5); } [[fallthrough]];
            case 6: if (6 < DEG) { PIPE_1B(6); } [[fallthrough]];
            case 7: if (7 < DEG) { PIPE_1B(7); } [[fallthrough]];
            case 8: if (8 < DEG) { PIPE_1B(8); } [[fallthrough]];
            case 9: if (9 < DEG) { PIPE_1B(9); } [[fallthrough]];
            case 10: if (10 < DEG) { PIPE_1B(10); } [[fallthrough]];
            case 11: if (11 < DEG) { PIPE_1B(11); } [[fallthrough]];
            case 12: if (12 < DEG) { PIPE_1B(12); } [[fallthrough]];
            case 13: if (13 < DEG) { PIPE_1B(13); } [[fallthrough]];
            case 14: if (14 < DEG) { PIPE_1B(14); } [[fallthrough]];
            case 15: if (15 < DEG) { PIPE_1B(15); } [[fallthrough]];
            case 16: if (16 < DEG) { PIPE_1B(16); } [[fallthrough]];
            case 17: if (17 < DEG) { PIPE_1B(17); } [[fallthrough]];
            case 18: if (18 < DEG) { PIPE_1B(18); } [[fallthrough]];
            case 19: if (19 < DEG) { PIPE_1B(19); } [[fallthrough]];
            case 20: if (20 < DEG) { PIPE_1B(20); } [[fallthrough]];
            case 21: if (21 < DEG) { PIPE_1B(21); } [[fallthrough]];
            case 22: if (22 < DEG) { PIPE_1B(22); } [[fallthrough]];
            case 23: if (23 < DEG) { PIPE_1B(23); } [[fallthrough]];
            case 24: if (24 < DEG) { PIPE_1B(24); } [[fallthrough]];
            case 25: if (25 < DEG) { PIPE_1B(25); } [[fallthrough]];
            case 26: if (26 < DEG) { PIPE_1B(26); } [[fallthrough]];
            default: break;
    }
}
template <int DEG>
__device__ __forceinline__ void pipe_1b_below(PipeBuf<DEG> &X, bool m0, int count)
{
    switch (count) {
            case 27: if (26 < DEG) { PIPE_1B(26); } [[fallthrough]];
            case 26: if (25 < DEG) { PIPE_1B(25); } [[fallthrough]];
            case 25: if (24 < DEG) { PIPE_1B(24); } [[fallthrough]];
            case 24: if (23 < DEG) { PIPE_1B(23); } [[fallthrough]];
            case 23: if (22 < DEG) { PIPE_1B(22); } [[fallthrough]];
            case 22: if (21 < DEG) { PIPE_1B(21); } [[fallthrough]];
            case 21: if (20 < DEG) { PIPE_1B(20); } [[fallthrough]];
            case 20: if (19 < DEG) { PIPE_1B(19); } [[fallthrough]];
            case 19: if (18 < DEG) { PIPE_1B(18); } [[fallthrough]];
            case 18: if (17 < DEG) { PIPE_1B(17); } [[fallthrough]];
            case 17: if (16 < DEG) { PIPE_1B(16); } [[fallthrough]];
            case 16: if (15 < DEG) { PIPE_1B(15); } [[fallthrough]];
            case 15: if (14 < DEG) { PIPE_1B(14); } [[fallthrough]];
            case 14: if (13 < DEG) { PIPE_1B(13); } [[fallthrough]];
            case 13: if (12 < DEG) { PIPE_1B(12); } [[fallthrough]];
            case 12: if (11 < DEG) { PIPE_1B(11); } [[fallthrough]];
            case 11: if (10 < DEG) { PIPE_1B(10); } [[fallthrough]];
            case 10: if (9 < DEG) { PIPE_1B(9); } [[fallthrough]];
            case 9: if (8 < DEG) { PIPE_1B(8); } [[fallthrough]];
            case 8: if (7 < DEG) { PIPE_1B(7); } [[fallthrough]];
            case 7: if (6 < DEG) { PIPE_1B(6); } [[fallthrough]];
            case 6: if (5 < DEG) { PIPE_1B(5); } [[fallthrough]];
            case 5: if (4 < DEG) { PIPE_1B(4); } [[fallthrough]];
            case 4: if (3 < DEG) { PIPE_1B(3); } [[fallthrough]];
            case 3: if (2 < DEG) { PIPE_1B(2); } [[fallthrough]];
            case 2: if (1 < DEG) { PIPE_1B(1); } [[fallthrough]];
            case 1: if (0 < DEG) { PIPE_1B(0); } [[fallthrough]];
            default: break;
    }
}

// finishes layer r held in X (pass 1b done for every slot) and prepares layer r+1 in Y
template <int DEG, int MODE>
__device__ __forceinline__ void pipe_layer(const FastCtx<MODE> &c, uint32_t fo, float (&nx)[3], bool act, int t, int r,
                                           PipeBuf<DEG> &X, PipeBuf<DEG> &Y)
{
    const int q = c.q;
    const const_u32 T = c.tab + r * LDPC_FAST_STRIDE;
    const uint32_t prim = T[27];
    const int ncf = (int)T[28];
    const bool more = r + 1 < q;
    const const_u32 Tn = c.tab + (more ? r + 1 : 0) * LDPC_FAST_STRIDE;
    const int ne = more ? (int)Tn[29] : 0;
    const bool mask0 = (r == 0) && (t == 0);
    float cst1 = 0.f, cst2 = 0.f;
    uint32_t tot = 0u, pkn = 0u;
    if (act) {
        cst1 = X.mn2 * c.alpha; cst2 = X.mn1 * c.alpha;
        tot = (uint32_t)(__popc(X.sgn) & 1);
        pkn = X.sgn ^ (tot ? ((1u << DEG) - 1u) : 0u);
        if (more) {
            // ---- layer r+1: addresses of every slot, loads of the early ones, packed state
            pipe_addr<DEG, MODE>(c, Tn, Y);
            pipe_loads_below<DEG, MODE>(c, fo, Y, ne);
            Y.c1o = nx[0]; Y.c2o = nx[1]; Y.pko = __float_as_uint(nx[2]);
            Y.mn1 = INFINITY; Y.mn2 = INFINITY; Y.sgn = 0u;
            const int rn = r + 2 < q ? r + 2 : 0;
            nx[0] = c.st_ld(fo, 0, rn); nx[1] = c.st_ld(fo, 1, rn); nx[2] = c.st_ld(fo, 2, rn);
        }
    }
    if (ncf > 0) __syncthreads();         // every read of layer r precedes its writes
    if (act) {
        // ---- pass 2 of layer r
        uint32_t idxn = 0u;
        const float m1s = __uint_as_float(__float_as_uint(cst1) | (tot << 31)), m2s = __uint_as_float(__float_as_uint(cst2) | (tot << 31));
#pragma unroll
        for (int j = 0; j < DEG; j++) {
            uint32_t off = ((prim >> j) & 1u) ? X.w[j] : c.redirect;
            if (j == DEG - 1 && mask0) off = c.redirect;
            const float x = X.v[j];
            const bool ismin = fabsf(x) == X.mn1;
            const float mag = ismin ? m1s : m2s;
            const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & 0x80000000u));
            idxn = ismin ? (uint32_t)j : idxn;
            c.post_st(off, fo, x + nw);
        }
        pkn |= idxn << 27;
        c.st_st(fo, 0, r, cst1); c.st_st(fo, 1, r, cst2); c.st_st(fo, 2, r, __uint_as_float(pkn));
    }
    // ---- duplicate edges of a bit-group inside layer r: ordered delta updates
    uint32_t prev_lvl = 0u;
    for (int i = 0; i < ncf; i++) {
        const uint32_t e = T[32 + i], meta = T[48 + i];
        const uint32_t j = meta & 31u, lvl = meta >> 8;
        if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
        if (act) {
            const uint32_t d = c.t4 - (e & 0x7FFu);
            const uint32_t off = min(d, d + (uint32_t)ROW_BYTES) + (e >> 11);
            const float nw = c2v_unpack_dyn<DEG>(cst1, cst2, pkn, j);
            const float od = c2v_unpack_dyn<DEG>(X.c1o, X.c2o, X.pko, j);
            const float L = c.post_ld(off, fo);
            c.post_st(off, fo, L + (nw - od));
        }
    }
    // ---- layer r+1, early slots: arithmetic while the stores of layer r drain
    if (act && more) pipe_1b_below<DEG>(Y, false, ne);
    __syncthreads();
    if (act && more) {
        pipe_loads_from<DEG, MODE>(c, fo, Y, ne);
        pipe_1b_from<DEG>(Y, false, ne);
    }
}

template <int DEG, int MODE>
__device__ __forceinline__ void fast_iteration_pipe(const FastCtx<MODE> &c, uint32_t fo, float (&nx)[3], bool act, int t)
{
    PipeBuf<DEG> A, B;
    const int q = c.q;
    // ---- prologue: layer 0 in full (the previous iteration ended with a barrier)
    if (act) {
        pipe_addr<DEG, MODE>(c, c.tab, A);
        pipe_loads_from<DEG, MODE>(c, fo, A, 0);
        A.c1o = nx[0]; A.c2o = nx[1]; A.pko = __float_as_uint(nx[2]);
        A.mn1 = INFINITY; A.mn2 = INFINITY; A.sgn = 0u;
        const int rn = 1 < q ? 1 : 0;
        nx[0] = c.st_ld(fo, 0, rn); nx[1] = c.st_ld(fo, 1, rn); nx[2] = c.st_ld(fo, 2, rn);
        pipe_1b_from<DEG>(A, t == 0, 0);
    }
    int r = 0;
    for (; r + 1 < q; r += 2) {
        pipe_layer<DEG, MODE>(c, fo, nx, act, t, r, A, B);
        pipe_layer<DEG, MODE>(c, fo, nx, act, t, r + 1, B, A);
    }
    if (r < q) pipe_layer<DEG, MODE>(c, fo, nx, act, t, r, A, B);
}
#undef PIPE_LD
#undef PIPE_1B

// syndrome of the hard decisions of one frame (enable_syndrome, depth 1): this lane's checks
template <int DEG, int MODE>
__device__ __forceinline__ int fast_syndrome(const FastCtx<MODE> &c, uint32_t fo, bool act, int t)
{
    int bad = 0;
    if (act)
        for (int r = 0; r < c.q; r++) {
            const const_u32 T = c.tab + r * LDPC_FAST_STRIDE;
            uint32_t x = 0u;
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const uint32_t e = T[j];
                const uint32_t d = c.t4 - (e & 0x7FFu);
                const uint32_t wo = min(d, d + (uint32_t)ROW_BYTES);
                float L;
                if (MODE >= 2) {
                    const uint32_t hb = (e >> 11) & 0x3FFFFu;
                    if (e & FE_LDS) L = c.lpost[(wo + hb) >> 2];
                    else L = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, wo, hb + fo, 0));
                } else L = c.post_ld(wo, (e >> 11) + fo);
                const bool absent = (j == DEG - 1) && (r == 0) && (t == 0);
                x ^= (!absent && L < 0.f) ? 1u : 0u;
            }
            bad |= (int)x;
        }
    return bad;
}

template <int DEG, int MODE, int NF>     // MODE 0: posteriors in LDS, 1: in the global workspace; NF frames per workgroup
__global__ void __launch_bounds__(LDPC_THREADS, NF == 2 ? 2 : 3)
ldpc_fast_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const bool act = t < LDPC_Z;
    const int q = p.q;
    float *gwork = p.gwork + (size_t)blockIdx.x * NF * p.gwork_words;
    FastCtx<MODE> c;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(gwork, 0, NF * p.gwork_words * 4, 0x00020000);   // wave-uniform by construction
    c.lpost = (lds_float *)smem;
    c.tab = (const_u32)p.fast_tab;
    c.t4 = (uint32_t)t * 4u;
    c.c2v_base = (uint32_t)p.glb_post_words * 4u;
    c.redirect = MODE == 0 ? (uint32_t)p.n_groups * ROW_BYTES + c.t4 : OOB;
    c.zero_row = (uint32_t)(p.lds_post_words - 2 * LDPC_Z) * 4u;      // MODE 2: last two rows of the LDS image
    c.junk_row = c.zero_row + ROW_BYTES;
    c.M = p.M; c.q = q; c.alpha = p.alpha;
    const const_u64 groups = (const_u64)p.groups;
    // where bit-group g lives: word offset (low) and LDS flag (high); modes 0 / 1 are uniform
    auto grp_ld = [&](int g, uint32_t idx4, uint32_t fo) -> float {
        if (MODE == 2) {
            const unsigned long long gl = groups[g];
            const uint32_t b = (uint32_t)gl * 4u;
            if (gl >> 32) return c.lpost[(b + idx4) >> 2];
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, idx4, b + fo, 0));
        }
        return c.post_ld(idx4, (uint32_t)g * ROW_BYTES + fo);
    };
    auto grp_st = [&](int g, uint32_t idx4, uint32_t fo, float v) {
        if (MODE == 2) {
            const unsigned long long gl = groups[g];
            const uint32_t b = (uint32_t)gl * 4u;
            if (gl >> 32) c.lpost[(b + idx4) >> 2] = v;
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), c.rs, idx4, b + fo, 0);
        } else c.post_st(idx4, (uint32_t)g * ROW_BYTES + fo, v);
    };
    if (MODE == 2) {   // the zero row is read by every edge that lives in global memory
        if (act) { c.lpost[(c.zero_row + c.t4) >> 2] = 0.f; c.lpost[(c.junk_row + c.t4) >> 2] = 0.f; }
        __syncthreads();
    }
    const uint32_t fstride = (uint32_t)p.gwork_words * 4u;      // bytes between the two frames' workspaces

    for (int f0 = blockIdx.x * NF; f0 < p.n_frames; f0 += gridDim.x * NF) {
        const int nfr = (p.n_frames - f0) < NF ? (p.n_frames - f0) : NF;
        // ---- channel LLRs -> posterior image; packed state := 0
        if (act)
            for (int k = 0; k < nfr; k++) {
                const float *Y = p.llr + (size_t)(f0 + k) * p.N;
                const uint32_t fo = (uint32_t)k * fstride;
                for (int g = 0; g < p.n_groups; g++) {
                    const int src = g < p.n_info ? g * LDPC_Z + t : p.K + q * t + (g - p.n_info);
                    grp_st(g, c.t4, fo, __builtin_nontemporal_load(&Y[src]));
                }
                for (int r = 0; r < q; r++) { c.st_st(fo, 0, r, 0.f); c.st_st(fo, 1, r, 0.f); c.st_st(fo, 2, r, 0.f); }
                if (p.inf_row >= 0) c.post_st(c.t4, (uint32_t)p.inf_row + fo, INFINITY);     // what NULL slots read
            }
        __syncthreads();

        int it[NF];
        bool ok[NF], live[NF];
#pragma unroll
        for (int k = 0; k < NF; k++) { it[k] = 0; ok[k] = false; live[k] = k < nfr; }
        // ---- two frames per lane while both are live
        if (NF == 2 && live[NF - 1]) {
            const uint32_t fo[2] = {0u, fstride};
            float nx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            while (it[0] < p.n_ite) {
                fast_iteration<DEG, MODE, 2>(c, fo, nx, act, t);
                it[0]++; it[NF - 1]++;
                if (p.early_stop || it[0] == p.n_ite) {
                    const int b0 = fast_syndrome<DEG, MODE>(c, fo[0], act, t), b1 = fast_syndrome<DEG, MODE>(c, fo[1], act, t);
                    ok[0] = !__syncthreads_or(b0);
                    ok[NF - 1] = !__syncthreads_or(b1);
                    if (ok[0] || ok[NF - 1]) break;
                }
            }
            live[0] = !ok[0] && it[0] < p.n_ite;
            live[NF - 1] = !ok[NF - 1] && it[NF - 1] < p.n_ite;
        }
        // ---- whatever is still live continues alone (odd tail, or its partner converged first)
#pragma unroll
        for (int k = 0; k < NF; k++) {
            if (!live[k]) continue;
            const uint32_t fo[1] = {(uint32_t)k * fstride};
            float nx[1][3] = {{0.f, 0.f, 0.f}};
            if (act) { nx[0][0] = c.st_ld(fo[0], 0, 0); nx[0][1] = c.st_ld(fo[0], 1, 0); nx[0][2] = c.st_ld(fo[0], 2, 0); }
            while (it[k] < p.n_ite) {
                fast_iteration<DEG, MODE, 1>(c, fo, nx, act, t);
                it[k]++;
                if (p.early_stop || it[k] == p.n_ite) {
                    ok[k] = !__syncthreads_or(fast_syndrome<DEG, MODE>(c, fo[0], act, t));
                    if (ok[k]) break;
                }
            }
        }

        // ---- outputs
        for (int k = 0; k < nfr; k++) {
            const int f = f0 + k;
            const uint32_t fo = (uint32_t)k * fstride;
            if (t == 0) {
                if (p.cwd) p.cwd[f] = ok[k] ? 1 : 0;
                if (p.ites) p.ites[f] = it[k];
            }
            if (act) {
                for (int g = 0; g < p.n_info; g++) {
                    const float L = grp_ld(g, c.t4, fo);
                    if (p.bits) __builtin_nontemporal_store((int32_t)(L < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + g * LDPC_Z + t]);
                    if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + t] = L;
                }
                if (p.post)
                    for (int g = p.n_info; g < p.n_groups; g++)
                        p.post[(size_t)f * p.N + p.K + q * t + (g - p.n_info)] = grp_ld(g, c.t4, fo);
            }
            if (p.packed) {
                // bit i of word w = info bit 32 w + i (tail bits zero)
                const int n_words = (p.K + 31) / 32;
                for (int wd = t; wd < n_words; wd += LDPC_THREADS) {
                    uint32_t word = 0u;
                    for (int b = 0; b < 32; b++) {
                        const int i = 32 * wd + b;
                        if (i >= p.K) break;
                        const int g = i / LDPC_Z;
                        const float L = grp_ld(g, (uint32_t)(i - g * LDPC_Z) * 4u, fo);
                        word |= (L < 0.f ? 1u : 0u) << b;
                    }
                    p.packed[(size_t)f * n_words + wd] = word;
                }
            }
        }
        __syncthreads();     // the posterior image is reused by the next frames of this workgroup
    }
}

// ---- sum-product (SPA, the reference's default --dec-implem): same schedule, same posterior image,
// but the c->v messages are kept per edge (fp32, [layer][slot][360] in the workspace) and the check
// node is the exact boxplus of the other v->c, by forward / backward recursions:
//     a [+] b = sign(a) sign(b) min(|a|,|b|) + log(1 + e^-|a+b|) - log(1 + e^-|a-b|)
// on the hardware exp2 / log2 units (absolute error ~1e-7 per operation).  +inf is the neutral
// element, which is also what absent / NULL slots carry.  Restated in the oracle (chk_update_spa).
__device__ __forceinline__ float boxplus(float a, float b)
{
    const float mn = fminf(fabsf(a), fabsf(b));
    const float sg = __uint_as_float(__float_as_uint(mn) | ((__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u));
    const float r = sg + (__logf(1.0f + __expf(-fabsf(a + b))) - __logf(1.0f + __expf(-fabsf(a - b))));
    return a == INFINITY ? b : (b == INFINITY ? a : r);
}

template <int DEG, int MODE>
__device__ __forceinline__ void fast_iteration_spa(const FastCtx<MODE> &c, bool act, int t)
{
    const int q = c.q;
    for (int r = 0; r < q; r++) {
        const const_u32 T = c.tab + r * LDPC_FAST_STRIDE;
        uint32_t E[DEG];
#pragma unroll
        for (int j = 0; j < DEG; j++) E[j] = T[j];
        const uint32_t prim = T[27];
        const int ncf = (int)T[28];
        const bool mask0 = (r == 0) && (t == 0);
        const uint32_t mbase = c.c2v_base + (uint32_t)(r * DEG) * ROW_BYTES;      // messages of this layer: [slot][360]
        float x[DEG], od[DEG], nw[DEG];
        uint32_t w[DEG];
        if (act) {
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const uint32_t d = c.t4 - (E[j] & 0x7FFu);
                w[j] = min(d, d + (uint32_t)ROW_BYTES) + (MODE == 0 ? (E[j] >> 11) : 0u);
                x[j] = c.post_ld(w[j], MODE == 0 ? 0u : (E[j] >> 11));
                od[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, c.t4, mbase + (uint32_t)j * ROW_BYTES, 0));
            }
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                x[j] = x[j] - od[j];
                if (j == DEG - 1 && mask0) x[j] = INFINITY;
            }
            float acc = INFINITY;
#pragma unroll
            for (int j = 0; j < DEG; j++) { nw[j] = acc; acc = boxplus(acc, x[j]); }           // forward: all before j
            acc = INFINITY;
#pragma unroll
            for (int j = DEG - 1; j >= 0; j--) { nw[j] = boxplus(nw[j], acc); acc = boxplus(acc, x[j]); }   // x backward: all after j
        }
        if (ncf > 0) __syncthreads();
        if (act) {
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                uint32_t off = ((prim >> j) & 1u) ? w[j] : c.redirect;
                if (j == DEG - 1 && mask0) off = c.redirect;
                c.post_st(off, MODE == 0 ? 0u : (E[j] >> 11), x[j] + nw[j]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, nw[j]), c.rs, c.t4, mbase + (uint32_t)j * ROW_BYTES, 0);
            }
        }
        // duplicate edges of a bit-group inside this layer: ordered delta updates, level by level
        uint32_t lvm[3] = {0u, 0u, 0u};
        for (int i = 0; i < ncf; i++) { const uint32_t meta = T[48 + i]; lvm[(meta >> 8) - 1u] |= 1u << (meta & 31u); }
        for (int lvl = 0; lvl < 3; lvl++) {
            if (!lvm[lvl]) break;
            __syncthreads();
            if (act) {
#pragma unroll
                for (int j = 0; j < DEG; j++)
                    if ((lvm[lvl] >> j) & 1u) {
                        const uint32_t so = MODE == 0 ? 0u : (E[j] >> 11);
                        const float L = c.post_ld(w[j], so);
                        c.post_st(w[j], so, L + (nw[j] - od[j]));
                    }
            }
        }
        __syncthreads();
    }
}

// ---- two frames per workgroup, one per HALF of a 12-wave workgroup.  A 6-wave workgroup cannot
// sit evenly on the CU's 4 SIMDs (2+2+1+1); two of them land 4+4+2+2 and the busiest SIMD then sets
// the pace: measured, one 6-wave workgroup per CU runs as fast per frame as two.  Twelve waves are
// dealt 3+3+3+3.  The halves share nothing but the barriers (both decode the same layer at the same
// time); each has its own workspace slot, LDS image, iteration count and early-stop decision.
template <int DEG, int MODE, bool PIPE, bool SPA = false>
__global__ void __launch_bounds__(2 * LDPC_THREADS, 3)
ldpc_fast2_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    __shared__ int s_flag[2];
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / LDPC_THREADS);     // wave-uniform, and provably so
    const int t = (int)threadIdx.x - half * LDPC_THREADS;
    const bool lane_ok = t < LDPC_Z;
    const int q = p.q;
    float *gwork = p.gwork + (size_t)(blockIdx.x * 2 + half) * p.gwork_words;
    FastCtx<MODE> c;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(gwork, 0, p.gwork_words * 4, 0x00020000);
    c.lpost = (lds_float *)smem + (size_t)half * p.lds_post_words;
    c.tab = (const_u32)p.fast_tab;
    c.t4 = (uint32_t)t * 4u;
    c.c2v_base = (uint32_t)p.glb_post_words * 4u;
    c.redirect = MODE == 0 ? (uint32_t)p.n_groups * ROW_BYTES + c.t4 : OOB;
    c.zero_row = 0u; c.junk_row = (uint32_t)(p.lds_post_words - LDPC_Z) * 4u;      // MODE 3: last row of the LDS image
    c.M = p.M; c.q = q; c.alpha = p.alpha;
    const uint32_t fo[1] = {0u};
    const const_u64 groups = (const_u64)p.groups;
    // where bit-group g lives (MODE 3: group table; modes 0 / 1: group g at row g)
    auto grp_ld = [&](int g, uint32_t idx4) -> float {
        if (MODE == 3) {
            const unsigned long long gl = groups[g];
            const uint32_t b = (uint32_t)gl * 4u;
            if (gl >> 32) return c.lpost[(b + idx4) >> 2];
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rs, idx4, b, 0));
        }
        return c.post_ld(idx4, (uint32_t)g * ROW_BYTES);
    };
    auto grp_st = [&](int g, uint32_t idx4, float v) {
        if (MODE == 3) {
            const unsigned long long gl = groups[g];
            const uint32_t b = (uint32_t)gl * 4u;
            if (gl >> 32) c.lpost[(b + idx4) >> 2] = v;
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), c.rs, idx4, b, 0);
        } else c.post_st(idx4, (uint32_t)g * ROW_BYTES, v);
    };

    for (int fb = blockIdx.x * 2; fb < p.n_frames; fb += gridDim.x * 2) {
        const int f = fb + half;
        const bool have = f < p.n_frames;
        const bool act = lane_ok && have;
        if (act) {
            const float *Y = p.llr + (size_t)f * p.N;
            for (int g = 0; g < p.n_groups; g++) {
                const int src = g < p.n_info ? g * LDPC_Z + t : p.K + q * t + (g - p.n_info);
                grp_st(g, c.t4, __builtin_nontemporal_load(&Y[src]));      // read-once stream: keep it out of the caches
            }
            if (SPA) { for (int e = 0; e < q * DEG; e++) __builtin_amdgcn_raw_buffer_store_b32(0u, c.rs, c.t4, c.c2v_base + (uint32_t)e * ROW_BYTES, 0); }
            else for (int r = 0; r < q; r++) { c.st_st(0u, 0, r, 0.f); c.st_st(0u, 1, r, 0.f); c.st_st(0u, 2, r, 0.f); }
            if (p.inf_row >= 0) c.post_st(c.t4, (uint32_t)p.inf_row, INFINITY);               // what NULL slots read
        }
        __syncthreads();
        int it = 0;
        bool ok = false, live = have;
        float nx[1][3] = {{0.f, 0.f, 0.f}};
        for (;;) {
            if (t == 0) s_flag[half] = 0;
            if (!__syncthreads_or(live ? 1 : 0)) break;          // also orders the flag reset
            if (SPA) fast_iteration_spa<DEG, MODE>(c, act && live, t);
            else if (PIPE) fast_iteration_pipe<DEG, MODE>(c, 0u, nx[0], act && live, t);
            else fast_iteration<DEG, MODE, 1>(c, fo, nx, act && live, t);
            int bad = 0;
            bool check = false;
            if (live) {
                it++;
                check = p.early_stop || it == p.n_ite;
                if (check) bad = fast_syndrome<DEG, MODE>(c, 0u, act, t);
            }
            if (bad) atomicOr(&s_flag[half], 1);
            __syncthreads();
            if (live && check) { ok = s_flag[half] == 0; if (ok || it == p.n_ite) live = false; }
            __syncthreads();                                      // flags are read before the next reset
        }
        if (have) {
            if (t == 0) {
                if (p.cwd) p.cwd[f] = ok ? 1 : 0;
                if (p.ites) p.ites[f] = it;
            }
            if (act) {
                for (int g = 0; g < p.n_info; g++) {
                    const float L = grp_ld(g, c.t4);
                    if (p.bits) __builtin_nontemporal_store((int32_t)(L < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + g * LDPC_Z + t]);
                    if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + t] = L;
                }
                if (p.post)
                    for (int g = p.n_info; g < p.n_groups; g++)
                        p.post[(size_t)f * p.N + p.K + q * t + (g - p.n_info)] = grp_ld(g, c.t4);
            }
            if (p.packed) {
                const int n_words = (p.K + 31) / 32;
                for (int wd = t; wd < n_words; wd += LDPC_THREADS) {
                    uint32_t word = 0u;
                    for (int b = 0; b < 32; b++) {
                        const int i = 32 * wd + b;
                        if (i >= p.K) break;
                        const int g = i / LDPC_Z;
                        word |= (grp_ld(g, (uint32_t)(i - g * LDPC_Z) * 4u) < 0.f ? 1u : 0u) << b;
                    }
                    p.packed[(size_t)f * n_words + wd] = word;
                }
            }
        }
        __syncthreads();
    }
}

template <int DEG, int MODE, bool PIPE, bool SPA = false>
static hipError_t fast2_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_fast2_kernel<DEG, MODE, PIPE, SPA>;
    static size_t configured_dev[64] = {0};
    int dev__ = 0;
    (void)hipGetDevice(&dev__);
    size_t &configured = configured_dev[dev__ & 63];
    const size_t lds = 2 * pl.lds_bytes;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    const int groups = (p.n_frames + 1) / 2;
    const int grid = groups < pl.grid_max ? groups : pl.grid_max;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * LDPC_THREADS), lds, s, p);
    return hipGetLastError();
}

template <int DEG, int MODE, bool PIPE, bool SPA = false>
static int fast2_occ(const LdpcPlan &pl)
{
    auto kern = ldpc_fast2_kernel<DEG, MODE, PIPE, SPA>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * pl.lds_bytes));
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 2 * LDPC_THREADS, 2 * pl.lds_bytes) != hipSuccess) nb = 1;
    if (const char *ev = getenv("DVBS2HIP_LDPC_BLOCKS_PER_CU")) { const int cap = atoi(ev); if (cap >= 1 && cap < nb) nb = cap; }
    return nb < 1 ? 1 : nb;
}

// ---- one frame per 6-wave workgroup, software-pipelined layers (up to 256 VGPRs)
template <int DEG, int MODE>
__global__ void __launch_bounds__(LDPC_THREADS, 2)
ldpc_fastp_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const bool act = t < LDPC_Z;
    const int q = p.q;
    float *gwork = p.gwork + (size_t)blockIdx.x * p.gwork_words;
    FastCtx<MODE> c;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(gwork, 0, p.gwork_words * 4, 0x00020000);
    c.lpost = (lds_float *)smem;
    c.tab = (const_u32)p.fast_tab;
    c.t4 = (uint32_t)t * 4u;
    c.c2v_base = (uint32_t)p.glb_post_words * 4u;
    c.redirect = MODE == 0 ? (uint32_t)p.n_groups * ROW_BYTES + c.t4 : OOB;
    c.zero_row = 0u; c.junk_row = 0u;
    c.M = p.M; c.q = q; c.alpha = p.alpha;
    for (int f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        if (act) {
            const float *Y = p.llr + (size_t)f * p.N;
            for (int g = 0; g < p.n_groups; g++) {
                const int src = g < p.n_info ? g * LDPC_Z + t : p.K + q * t + (g - p.n_info);
                c.post_st(c.t4, (uint32_t)g * ROW_BYTES, __builtin_nontemporal_load(&Y[src]));      // read-once stream: keep it out of the caches
            }
            for (int r = 0; r < q; r++) { c.st_st(0u, 0, r, 0.f); c.st_st(0u, 1, r, 0.f); c.st_st(0u, 2, r, 0.f); }
            if (p.inf_row >= 0) c.post_st(c.t4, (uint32_t)p.inf_row, INFINITY);               // what NULL slots read
        }
        __syncthreads();
        int it = 0;
        bool ok = false;
        float nx[3] = {0.f, 0.f, 0.f};
        while (it < p.n_ite) {
            fast_iteration_pipe<DEG, MODE>(c, 0u, nx, act, t);
            it++;
            if (p.early_stop || it == p.n_ite) {
                ok = !__syncthreads_or(fast_syndrome<DEG, MODE>(c, 0u, act, t));
                if (ok) break;
            }
        }
        if (t == 0) {
            if (p.cwd) p.cwd[f] = ok ? 1 : 0;
            if (p.ites) p.ites[f] = it;
        }
        if (act) {
            for (int g = 0; g < p.n_info; g++) {
                const float L = c.post_ld(c.t4, (uint32_t)g * ROW_BYTES);
                if (p.bits) __builtin_nontemporal_store((int32_t)(L < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + g * LDPC_Z + t]);
                if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + t] = L;
            }
            if (p.post)
                for (int g = p.n_info; g < p.n_groups; g++)
                    p.post[(size_t)f * p.N + p.K + q * t + (g - p.n_info)] = c.post_ld(c.t4, (uint32_t)g * ROW_BYTES);
        }
        if (p.packed) {
            const int n_words = (p.K + 31) / 32;
            for (int wd = t; wd < n_words; wd += LDPC_THREADS) {
                uint32_t word = 0u;
                for (int b = 0; b < 32; b++) {
                    const int i = 32 * wd + b;
                    if (i >= p.K) break;
                    word |= (c.post_ld((uint32_t)i * 4u, 0u) < 0.f ? 1u : 0u) << b;
                }
                p.packed[(size_t)f * n_words + wd] = word;
            }
        }
        __syncthreads();
    }
}

template <int DEG, int MODE>
static hipError_t fastp_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_fastp_kernel<DEG, MODE>;
    static size_t configured_dev[64] = {0};
    int dev__ = 0;
    (void)hipGetDevice(&dev__);
    size_t &configured = configured_dev[dev__ & 63];
    if (pl.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        configured = pl.lds_bytes;
    }
    const int grid = p.n_frames < pl.grid_max ? p.n_frames : pl.grid_max;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_THREADS), pl.lds_bytes, s, p);
    return hipGetLastError();
}
template <int DEG, int MODE>
static int fastp_occ(const LdpcPlan &pl)
{
    auto kern = ldpc_fastp_kernel<DEG, MODE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, LDPC_THREADS, pl.lds_bytes) != hipSuccess) nb = 1;
    if (const char *ev = getenv("DVBS2HIP_LDPC_BLOCKS_PER_CU")) { const int cap = atoi(ev); if (cap >= 1 && cap < nb) nb = cap; }
    return nb < 1 ? 1 : nb;
}
#define FASTP_DISPATCH(FN, ...)                                                                   \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0>(__VA_ARGS__) : FN<27, 1>(__VA_ARGS__))     \
                      : (pl.fast_mode == 0 ? FN<11, 0>(__VA_ARGS__) : FN<11, 1>(__VA_ARGS__)))

template <int DEG, int MODE, int NF>
static hipError_t fast_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_fast_kernel<DEG, MODE, NF>;
    static size_t configured_dev[64] = {0};
    int dev__ = 0;
    (void)hipGetDevice(&dev__);
    size_t &configured = configured_dev[dev__ & 63];
    if (pl.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        configured = pl.lds_bytes;
    }
    const int groups = (p.n_frames + NF - 1) / NF;
    const int grid = groups < pl.grid_max ? groups : pl.grid_max;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_THREADS), pl.lds_bytes, s, p);
    return hipGetLastError();
}

template <int DEG, int MODE, int NF>
static int fast_occ(const LdpcPlan &pl)
{
    auto kern = ldpc_fast_kernel<DEG, MODE, NF>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, LDPC_THREADS, pl.lds_bytes) != hipSuccess) nb = 1;
    if (const char *ev = getenv("DVBS2HIP_LDPC_BLOCKS_PER_CU")) { const int cap = atoi(ev); if (cap >= 1 && cap < nb) nb = cap; }
    return nb < 1 ? 1 : nb;
}

#define FAST_DISPATCH_D(D, FN, ...)                                                             \
    (pl.fast_mode == 0 ? FN<D, 0, 1>(__VA_ARGS__)                                                \
     : pl.fast_mode == 2 ? FN<D, 2, 1>(__VA_ARGS__)                                              \
     : (pl.fast_nf == 2 ? FN<D, 1, 2>(__VA_ARGS__) : FN<D, 1, 1>(__VA_ARGS__)))
#define FAST_DISPATCH(FN, ...) (pl.fast_deg == 27 ? FAST_DISPATCH_D(27, FN, __VA_ARGS__) : pl.fast_deg == 13 ? FAST_DISPATCH_D(13, FN, __VA_ARGS__) : FAST_DISPATCH_D(11, FN, __VA_ARGS__))

// (the pipelined iteration needs ~230 VGPRs at DEG 27: more than the 168 a 12-wave workgroup may use)
#define FAST2_DISPATCH(FN, ...)                                                                   \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0, false>(__VA_ARGS__) : pl.fast_mode == 3 ? FN<27, 3, false>(__VA_ARGS__) : FN<27, 1, false>(__VA_ARGS__))      \
     : pl.fast_deg == 13 ? (pl.fast_mode == 0 ? FN<13, 0, false>(__VA_ARGS__) : FN<13, 1, false>(__VA_ARGS__))     \
                      : (pl.fast_pipe ? (pl.fast_mode == 0 ? FN<11, 0, true>(__VA_ARGS__) : FN<11, 1, true>(__VA_ARGS__)) \
                                      : (pl.fast_mode == 0 ? FN<11, 0, false>(__VA_ARGS__) : FN<11, 1, false>(__VA_ARGS__))))

#define FASTSPA_DISPATCH(FN, ...)                                                                                  \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0, false, true>(__VA_ARGS__) : FN<27, 1, false, true>(__VA_ARGS__))   \
     : pl.fast_deg == 13 ? (pl.fast_mode == 0 ? FN<13, 0, false, true>(__VA_ARGS__) : FN<13, 1, false, true>(__VA_ARGS__)) \
                         : (pl.fast_mode == 0 ? FN<11, 0, false, true>(__VA_ARGS__) : FN<11, 1, false, true>(__VA_ARGS__)))

int ldpc_fast_blocks_per_cu(const LdpcPlan &pl)
{
    if (pl.spa) return FASTSPA_DISPATCH(fast2_occ, pl);
    if (pl.fast_wf == 2) return FAST2_DISPATCH(fast2_occ, pl);
    if (pl.fast_pipe && pl.fast_mode != 2 && pl.fast_nf == 1 && pl.fast_deg != 13) return FASTP_DISPATCH(fastp_occ, pl);
    return FAST_DISPATCH(fast_occ, pl);
}

hipError_t ldpc_fast_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    p.fast_tab = pl.d_fast_tab; p.groups = pl.d_groups; p.pipe = pl.fast_pipe ? 1 : 0; p.inf_row = pl.fast_inf_row;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    p.lds_post_words = pl.lds_post_words; p.glb_post_words = pl.glb_post_words; p.gwork_words = pl.gwork_words;
    if (pl.spa) return FASTSPA_DISPATCH(fast2_inst, pl, p, s);
    if (pl.fast_wf == 2) return FAST2_DISPATCH(fast2_inst, pl, p, s);
    if (pl.fast_pipe && pl.fast_mode != 2 && pl.fast_nf == 1 && pl.fast_deg != 13) return FASTP_DISPATCH(fastp_inst, pl, p, s);
    return FAST_DISPATCH(fast_inst, pl, p, s);
}

}  // namespace dvbs2
