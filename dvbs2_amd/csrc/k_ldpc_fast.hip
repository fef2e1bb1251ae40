// a1, the earlier fast path -- layered NMS / SPA for codes whose layers fit 27 slots, TWO frames per
// 12-wave workgroup (one per half, 3 waves on every SIMD), same schedule and arithmetic as the oracle
// (bit-exact for NMS).  k_ldpc_wg8.hip is the production NMS kernel; this file carries
//   * the sum-product check node (`--dec-implem SPA`, the reference's default), and
//   * the NMS kernel the round started from (DVBS2HIP_LDPC_WG=12, and the fall-back when the plan cannot give
//     k_ldpc_wg8 its image), kept as the measured reference point of DESIGN.md section 6.
// What it is built around on gfx950:
//   * the CU's single scalar unit: a layer costs ~2 SALU per edge -- the layer table is one dword per slot
//     (byte shift | byte offset of the bit-group), preloaded with wide s_loads; no per-edge branch: the few
//     same-layer duplicate edges are redirected with a select in pass 2 and replayed from a short list;
//   * address VALU: per-lane byte offsets w_j = ((t - t0_j) mod 360) * 4 computed once per layer and kept
//     for the store pass; global posteriors through ONE raw buffer descriptor (voffset + SGPR soffset);
//   * latency: all posterior loads of a check are issued before the first is used; the next layer's packed
//     c->v state (private to the lane) is prefetched under the compute.
// Posterior image: bit-group g at word 360 g (info groups, then parity groups regrouped [r][t]), in LDS
// (MODE 0, N = 16200), in the workgroup's global slot (MODE 1) or split at compile time (MODE 3, below).
// Packed c->v state always in the global slot: 12 B per check per layer, coalesced, prefetched.
#include "dvbs2hip_internal.h"
#include <cstdlib>

namespace dvbs2 {

typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(4))) uint32_t *const_u32;
typedef const __attribute__((address_space(4))) unsigned long long *const_u64;
// MODE 3 = STATIC hybrid: the host picks the LDS-resident bit-groups so that EVERY layer has exactly
// NL_STATIC of its 27 slots in LDS and sorts them first: slot j < NL_STATIC is an LDS access, the
// others are buffer accesses -- decided at compile time, no per-slot branch, select or dual issue.
constexpr int NL_STATIC = 9;
constexpr uint32_t FE_LDS = 1u << 29;       // table entry: the bit-group lives in LDS (static hybrid image)
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
    uint32_t junk_row;           // MODE 3: byte offset of the write-only LDS row
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
                w[j] = min(d, d + (uint32_t)ROW_BYTES) + (MODE == 0 ? (E[j] >> 11) : 0u);   // LDS: full address
#pragma unroll
                for (int k = 0; k < NA; k++) v[k][j] = c.slot_ld(j, w[j], E[j], fo[k]);
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
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    const float x = v[k][j];
                    const bool ismin = fabsf(x) == mn1[k];
                    const float mag = ismin ? m1s[k] : m2s[k];
                    const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & 0x80000000u));
                    idxn[k] = ismin ? (uint32_t)j : idxn[k];
                    if (MODE == 3) c.slot_st(j, w[j], keep, E[j], fo[k], x + nw);
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
                    if (MODE == 3) {
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
                if (MODE == 3) {
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
    const float r = sg + (hw_log(1.0f + hw_exp(-fabsf(a + b))) - hw_log(1.0f + hw_exp(-fabsf(a - b))));
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
template <int DEG, int MODE, bool SPA = false>
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
    c.junk_row = (uint32_t)(p.lds_post_words - LDPC_Z) * 4u;      // MODE 3: last row of the LDS image
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

template <int DEG, int MODE, bool SPA = false>
static hipError_t fast2_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_fast2_kernel<DEG, MODE, SPA>;
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

template <int DEG, int MODE, bool SPA = false>
static int fast2_occ(const LdpcPlan &pl)
{
    auto kern = ldpc_fast2_kernel<DEG, MODE, SPA>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * pl.lds_bytes));
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 2 * LDPC_THREADS, 2 * pl.lds_bytes) != hipSuccess) nb = 1;
    if (const char *ev = getenv("DVBS2HIP_LDPC_BLOCKS_PER_CU")) { const int cap = atoi(ev); if (cap >= 1 && cap < nb) nb = cap; }
    return nb < 1 ? 1 : nb;
}

#define FAST2_DISPATCH(FN, ...)                                                                                                                          \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0>(__VA_ARGS__) : pl.fast_mode == 3 ? FN<27, 3>(__VA_ARGS__) : FN<27, 1>(__VA_ARGS__))            \
     : pl.fast_deg == 13 ? (pl.fast_mode == 0 ? FN<13, 0>(__VA_ARGS__) : FN<13, 1>(__VA_ARGS__))                                                      \
                         : (pl.fast_mode == 0 ? FN<11, 0>(__VA_ARGS__) : FN<11, 1>(__VA_ARGS__)))
#define FASTSPA_DISPATCH(FN, ...)                                                                                  \
    (pl.fast_deg == 27 ? (pl.fast_mode == 0 ? FN<27, 0, true>(__VA_ARGS__) : FN<27, 1, true>(__VA_ARGS__))         \
     : pl.fast_deg == 13 ? (pl.fast_mode == 0 ? FN<13, 0, true>(__VA_ARGS__) : FN<13, 1, true>(__VA_ARGS__))       \
                         : (pl.fast_mode == 0 ? FN<11, 0, true>(__VA_ARGS__) : FN<11, 1, true>(__VA_ARGS__)))

int ldpc_fast_blocks_per_cu(const LdpcPlan &pl)
{
    if (pl.spa) return FASTSPA_DISPATCH(fast2_occ, pl);
    return FAST2_DISPATCH(fast2_occ, pl);
}

hipError_t ldpc_fast_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    p.fast_tab = pl.d_fast_tab; p.groups = pl.d_groups; p.inf_row = pl.fast_inf_row;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    p.lds_post_words = pl.lds_post_words; p.glb_post_words = pl.glb_post_words; p.gwork_words = pl.gwork_words;
    if (pl.spa) return FASTSPA_DISPATCH(fast2_inst, pl, p, s);
    return FAST2_DISPATCH(fast2_inst, pl, p, s);
}

}  // namespace dvbs2
