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
//     next layer's packed c->v state (private to the lane) is prefetched under the compute.
//
// Posterior image: bit-group g at word 360 g (info groups, then parity groups regrouped
// [r][t]), either entirely in LDS (N = 16200: 64.8 KB, two workgroups per CU) or entirely
// in the workgroup's global workspace slot (N = 64800), which is reused frame after frame
// and therefore stays L2 / Infinity-Cache resident.  Packed c->v state always lives in the
// workspace: 12 B per check per layer, coalesced, prefetched.
#include "dvbs2hip_internal.h"

namespace dvbs2 {

typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(4))) uint32_t *const_u32;

constexpr uint32_t OOB = 0x7FFFF000u;       // beyond every workspace: loads return 0, stores are dropped
constexpr int ROW_BYTES = LDPC_Z * 4;

__device__ __forceinline__ float c2v_unpack_dyn(float c1, float c2, uint32_t pk, uint32_t j)
{
    const float mag = ((pk >> 27) == j) ? c1 : c2;
    return __uint_as_float(__float_as_uint(mag) | ((pk << (31u - j)) & 0x80000000u));
}

template <int DEG, int MODE>     // MODE 0: posteriors in LDS, 1: posteriors in the global workspace
__global__ void __launch_bounds__(LDPC_THREADS, 3)
ldpc_fast_kernel(const LdpcKParams p)
{
    extern __shared__ float smem[];
    lds_float *lpost = (lds_float *)smem;
    const int t = threadIdx.x;
    const bool act = t < LDPC_Z;
    const int M = p.M, q = p.q;
    const const_u32 tab = (const_u32)p.fast_tab;
    const uint32_t t4 = (uint32_t)t * 4u;
    const uint32_t c2v_base = (uint32_t)p.glb_post_words * 4u;         // byte offset of the packed state
    const uint32_t dummy = (uint32_t)p.n_groups * ROW_BYTES + t4;       // LDS dummy row (mode 0)
    const uint32_t redirect = MODE == 0 ? dummy : OOB;

    // one descriptor for the workgroup's whole workspace slot (wave-uniform by construction)
    float *gwork = p.gwork + (size_t)blockIdx.x * p.gwork_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gwork, 0, p.gwork_words * 4, 0x00020000);

#define POST_LD(off, soff) (MODE == 0 ? lpost[((off) + (soff)) >> 2] : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (off), (soff), 0)))
#define POST_ST(off, soff, val) do { if (MODE == 0) lpost[((off) + (soff)) >> 2] = (val); \
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)(val)), rs, (off), (soff), 0); } while (0)
#define ST_LD(arr, r_) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, t4, c2v_base + (uint32_t)((arr) * M + (r_) * LDPC_Z) * 4u, 0))
#define ST_ST(arr, r_, val) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)(val)), rs, t4, c2v_base + (uint32_t)((arr) * M + (r_) * LDPC_Z) * 4u, 0)

    for (int f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        const float *Y = p.llr + (size_t)f * p.N;
        // ---- channel LLRs -> posterior image; packed state := 0
        if (act) {
            for (int g = 0; g < p.n_groups; g++) {
                const int src = g < p.n_info ? g * LDPC_Z + t : p.K + q * t + (g - p.n_info);
                POST_ST(t4, (uint32_t)g * ROW_BYTES, Y[src]);
            }
            for (int r = 0; r < q; r++) { ST_ST(0, r, 0.f); ST_ST(1, r, 0.f); ST_ST(2, r, 0.f); }
        }
        __syncthreads();

        int it = 0;
        bool ok = false;
        float nx1 = 0.f, nx2 = 0.f, nxk = 0.f;
        while (it < p.n_ite) {
            for (int r = 0; r < q; r++) {
                const const_u32 T = tab + r * LDPC_FAST_STRIDE;
                uint32_t E[DEG];
#pragma unroll
                for (int j = 0; j < DEG; j++) E[j] = T[j];
                const uint32_t prim = T[27];
                const int ncf = (int)T[28];
                const bool mask0 = (r == 0) && (t == 0);        // p_{c-1} of check 0 does not exist
                float v[DEG];
                uint32_t w[DEG];
                float cst1 = 0.f, cst2 = 0.f, mn1 = INFINITY, mn2 = INFINITY;
                const float c1o = nx1, c2o = nx2;
                const uint32_t pko = __float_as_uint(nxk);
                uint32_t sacc = 0u, pkn = 0u, idxn = 0u;
                if (act) {
                    // ---- pass 1a: every posterior load of the check in flight before any use
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        const uint32_t d = t4 - (E[j] & 0x7FFu);
                        w[j] = min(d, d + (uint32_t)ROW_BYTES) + (MODE == 0 ? (E[j] >> 11) : 0u);   // LDS: full address
                        v[j] = POST_LD(w[j], MODE == 0 ? 0u : (E[j] >> 11));
                    }
                    {   // prefetch the next layer's packed state (private to this lane)
                        const int rn = r + 1 < q ? r + 1 : 0;
                        nx1 = ST_LD(0, rn); nx2 = ST_LD(1, rn); nxk = ST_LD(2, rn);
                    }
                    // ---- pass 1b: v->c = posterior - old c->v ; running min1 / min2 / sign
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        float x = v[j] - c2v_unpack_dyn(c1o, c2o, pko, (uint32_t)j);
                        if (j == DEG - 1 && mask0) x = INFINITY;
                        v[j] = x;
                        const float a = fabsf(x);
                        mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                        mn1 = fminf(mn1, a);
                        sacc ^= __float_as_uint(x);
                    }
                    cst1 = mn2 * p.alpha;
                    cst2 = mn1 * p.alpha;
                }
                if (ncf > 0) __syncthreads();         // every read of the layer precedes its writes
                if (act) {
                    // ---- pass 2: new c->v ; posterior = v->c + new c->v.  Duplicate edges (not in
                    //      `prim`) and the absent edge are redirected, not branched around.
#pragma unroll
                    for (int j = 0; j < DEG; j++) {
                        const float x = v[j];
                        const bool ismin = fabsf(x) == mn1;
                        const float mag = ismin ? cst1 : cst2;
                        const uint32_t s = (sacc ^ __float_as_uint(x)) & 0x80000000u;
                        const float nw = __uint_as_float(__float_as_uint(mag) | s);
                        pkn |= s >> (31 - j);
                        idxn = ismin ? (uint32_t)j : idxn;
                        uint32_t off = ((prim >> j) & 1u) ? w[j] : redirect;
                        if (j == DEG - 1 && mask0) off = redirect;
                        POST_ST(off, MODE == 0 ? 0u : (E[j] >> 11), x + nw);
                    }
                    pkn |= idxn << 27;
                    ST_ST(0, r, cst1); ST_ST(1, r, cst2); ST_ST(2, r, __uint_as_float(pkn));
                    if (q == 1) { nx1 = cst1; nx2 = cst2; nxk = __uint_as_float(pkn); }
                }
                // ---- duplicate edges of a bit-group inside this layer: ordered delta updates
                uint32_t prev_lvl = 0u;
                for (int k = 0; k < ncf; k++) {
                    const uint32_t e = T[32 + k], meta = T[48 + k];
                    const uint32_t j = meta & 31u, lvl = meta >> 8;
                    if (lvl != prev_lvl) { __syncthreads(); prev_lvl = lvl; }
                    if (act) {
                        const uint32_t d = t4 - (e & 0x7FFu);
                        const uint32_t off = min(d, d + (uint32_t)ROW_BYTES);
                        const float nw = c2v_unpack_dyn(cst1, cst2, pkn, j);
                        const float od = c2v_unpack_dyn(c1o, c2o, pko, j);
                        const float L = POST_LD(off, e >> 11);
                        POST_ST(off, e >> 11, L + (nw - od));
                    }
                }
                __syncthreads();
            }
            it++;
            if (p.early_stop || it == p.n_ite) {
                // ---- syndrome of the hard decisions (enable_syndrome, depth 1)
                int bad = 0;
                if (act)
                    for (int r = 0; r < q; r++) {
                        const const_u32 T = tab + r * LDPC_FAST_STRIDE;
                        uint32_t x = 0u;
#pragma unroll
                        for (int j = 0; j < DEG; j++) {
                            const uint32_t e = T[j];
                            const uint32_t d = t4 - (e & 0x7FFu);
                            const float L = POST_LD(min(d, d + (uint32_t)ROW_BYTES), e >> 11);
                            const bool absent = (j == DEG - 1) && (r == 0) && (t == 0);
                            x ^= (!absent && L < 0.f) ? 1u : 0u;
                        }
                        bad |= (int)x;
                    }
                ok = !__syncthreads_or(bad);
                if (ok) break;
            }
        }

        // ---- outputs
        if (t == 0) {
            if (p.cwd) p.cwd[f] = ok ? 1 : 0;
            if (p.ites) p.ites[f] = it;
        }
        if (act) {
            for (int g = 0; g < p.n_info; g++) {
                const float L = POST_LD(t4, (uint32_t)g * ROW_BYTES);
                if (p.bits) p.bits[(size_t)f * p.K + g * LDPC_Z + t] = L < 0.f ? 1 : 0;
                if (p.post) p.post[(size_t)f * p.N + g * LDPC_Z + t] = L;
            }
            if (p.post)
                for (int g = p.n_info; g < p.n_groups; g++)
                    p.post[(size_t)f * p.N + p.K + q * t + (g - p.n_info)] = POST_LD(t4, (uint32_t)g * ROW_BYTES);
        }
        if (p.packed) {
            // bit i of word w = info bit 32 w + i (tail bits zero)
            const int n_words = (p.K + 31) / 32;
            for (int wd = t; wd < n_words; wd += LDPC_THREADS) {
                uint32_t word = 0u;
                for (int b = 0; b < 32; b++) {
                    const int k = 32 * wd + b;
                    if (k >= p.K) break;
                    const float L = POST_LD((uint32_t)k * 4u, 0u);
                    word |= (L < 0.f ? 1u : 0u) << b;
                }
                p.packed[(size_t)f * n_words + wd] = word;
            }
        }
        __syncthreads();     // the posterior image is reused by the next frame of this workgroup
    }
#undef POST_LD
#undef POST_ST
#undef ST_LD
#undef ST_ST
}

template <int DEG, int MODE>
static hipError_t fast_inst(const LdpcPlan &pl, const LdpcKParams &p, hipStream_t s)
{
    auto kern = ldpc_fast_kernel<DEG, MODE>;
    static size_t configured = 0;
    if (pl.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        configured = pl.lds_bytes;
    }
    const int grid = p.n_frames < pl.grid_max ? p.n_frames : pl.grid_max;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(LDPC_THREADS), pl.lds_bytes, s, p);
    return hipGetLastError();
}

template <int DEG, int MODE>
static int fast_occ(const LdpcPlan &pl)
{
    auto kern = ldpc_fast_kernel<DEG, MODE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, LDPC_THREADS, pl.lds_bytes) != hipSuccess) nb = 1;
    return nb < 1 ? 1 : nb;
}

#define FAST_DISPATCH(FN, ...)                                                                  \
    (pl.deg_max == 27 ? (pl.fast_mode == 0 ? FN<27, 0>(__VA_ARGS__) : FN<27, 1>(__VA_ARGS__))   \
                      : (pl.fast_mode == 0 ? FN<11, 0>(__VA_ARGS__) : FN<11, 1>(__VA_ARGS__)))

int ldpc_fast_blocks_per_cu(const LdpcPlan &pl) { return FAST_DISPATCH(fast_occ, pl); }

hipError_t ldpc_fast_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    p.fast_tab = pl.d_fast_tab;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    p.lds_post_words = pl.lds_post_words; p.glb_post_words = pl.glb_post_words; p.gwork_words = pl.gwork_words;
    return FAST_DISPATCH(fast_inst, pl, p, s);
}

}  // namespace dvbs2
