// a1 for SMALL batches of short frames (BASELINE configs[4], "small-frame latency"; round 6): one frame per CU, TWO LANES PER CHECK.
// OPT-IN (DVBS2HIP_LDPC_LAT=1): bit-exact against the oracle on the first run, and 19-26 % SLOWER than what it was written to beat -- kept as the measured record of that, and as
// the shortest statement of the QC-layer schedule on the device (200 lines, no tuning knobs).
//
// k_ldpc_wg8.hip is shaped for throughput: two frames per CU, six working waves per frame, one lane per check.  A call that brings at most one frame per CU leaves that
// shape with a lone workgroup whose layer is a chain of LDS round trips and ~500 dependent vector instructions per lane on four SIMDs that hold one or two waves each
// (results/r06/phase_f1_qpsk_s.txt: 5 600 cycles per layer of the short 8/9 code).  Here a check's slots are split over two ADJACENT lanes -- lane 2 t takes the first
// half of check t's slots, lane 2 t + 1 the second -- so a frame is twelve waves, three per SIMD, each with half the work per layer; the halves merge {min1, min2, signs,
// position} with three DPP swaps inside their lane pair (no LDS, no barrier: what k_ldpc_nat.hip's part kernels do).  The posterior image (N = 16200: 64.8 KB) AND the
// packed c->v state (12 bytes per check) live in LDS; nothing of a frame is in global memory between its input and its output.
//
// Same schedule, same arithmetic as the QC-layer kernels and the oracle's ORC_SCHED_QC: every check of a layer reads before any writes (a barrier, only in layers with
// duplicate edges), primary edges write v->c + new, duplicate edges add (new - old) level by level.  min1 / min2 of the two halves' union = the scan's (order-free), the
// value-equality rule of AFF3CT's min-sum, alpha applied to the two minima: posteriors are BIT-IDENTICAL (tests/test_ldpc_gpu.py).
// Tables: the LDS-only plan of k_ldpc.hip as it is (w8_tab: slot entries = byte shift | byte offset of the bit-group row << 11, prim mask, conflict list).
// Min-sum / normalised min-sum only; used by dvbs2hip_api.hip for calls of at most one frame per CU on codes whose image and state fit the LDS.
#include "dvbs2hip_internal.h"

namespace dvbs2 {

typedef __attribute__((address_space(3))) float lat_lds_float;
typedef __attribute__((address_space(3))) uint32_t lat_lds_u32;
typedef const __attribute__((address_space(4))) uint32_t *lat_const_u32;
constexpr int LAT_THREADS = 768, LAT_ROW = LDPC_Z * 4;

__device__ __forceinline__ lat_lds_float *lat_f(uint32_t a) { return (lat_lds_float *)(size_t)a; }
__device__ __forceinline__ lat_lds_u32 *lat_u(uint32_t a) { return (lat_lds_u32 *)(size_t)a; }
__device__ __forceinline__ float lat_swap(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }      // quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t lat_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }

template <int DEG>
__global__ void __launch_bounds__(LAT_THREADS)
ldpc_lat_kernel(const LdpcKParams p)
{
    extern __shared__ float lat_smem[];
    if ((uint32_t)(size_t)(lat_lds_float *)lat_smem != 0u) __builtin_trap();      // LDS is addressed by plain byte offsets
    constexpr int H = (DEG + 1) / 2;                       // slots of the first half-check; the second holds DEG - H
    const int L = (int)threadIdx.x, t = L >> 1, part = L & 1;
    const bool act = t < LDPC_Z;
    const uint32_t t4 = (uint32_t)(act ? t : 0) * 4u;
    const int q = p.q, N = p.N, K = p.K, M = p.M, n_info = p.n_info;
    const lat_const_u32 tab = (lat_const_u32)p.w8.tab;
    const uint32_t junk = p.w8.lds_junk, inf_row = junk + (uint32_t)LAT_ROW;
    const uint32_t st0 = inf_row + (uint32_t)LAT_ROW;      // packed state: [layer][check][3] dwords
    const uint32_t flag_a = st0 + (uint32_t)M * 12u;       // one word: "some check of the sweep is unsatisfied"
    const uint32_t SB = 0x80000000u;

    for (int f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        // ---- channel LLRs -> posterior image (information bit i at byte 4 i; parity bit c at row n_info + c mod q, element c / q), state := 0
        const float *Y = p.llr + (size_t)f * N;
        for (int i = L; i < N; i += LAT_THREADS) {
            const float y = __builtin_nontemporal_load(Y + i);
            uint32_t a;
            if (i < K) a = (uint32_t)i * 4u;
            else { const int c = i - K, tt = c / q, r = c - tt * q; a = (uint32_t)((n_info + r) * LDPC_Z + tt) * 4u; }
            *lat_f(a) = y;
        }
        for (int i = L; i < LDPC_Z; i += LAT_THREADS) { *lat_f(inf_row + (uint32_t)i * 4u) = INFINITY; *lat_f(junk + (uint32_t)i * 4u) = 0.f; }
        for (int i = L; i < 3 * M; i += LAT_THREADS) *lat_u(st0 + (uint32_t)i * 4u) = 0u;
        if (L == 0) *lat_u(flag_a) = 0u;
        __syncthreads();

        int it = 0;
        bool ok = false;
        while (it < p.n_ite) {
            for (int r = 0; r < q; r++) {
                const lat_const_u32 T = tab + r * LDPC_FAST_STRIDE;
                const uint32_t prim = T[27], ncf = T[28] & 0xFFu;
                // ---- this half-check's slots: posterior loads, v->c = posterior - old c->v, local minima and signs
                const uint32_t sa = st0 + (uint32_t)(r * LDPC_Z + (act ? t : 0)) * 12u;
                const float c1o = *lat_f(sa), c2o = *lat_f(sa + 4u);
                const uint32_t pko = *lat_u(sa + 8u), idxo = pko >> 27;
                float x[H], old[H];
                uint32_t adr[H];
                float mn1 = INFINITY, mn2 = INFINITY;
                uint32_t sw = 0u;
                int li = -1;
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const bool has2 = H + i < DEG;                                   // (compile time: the second half may be one slot shorter)
                    const uint32_t e = part ? (has2 ? T[H + i < DEG ? H + i : 0] : 0u) : T[i];
                    const bool valid = act && (part == 0 || has2);
                    const uint32_t js = (uint32_t)(part ? H + i : i);
                    const uint32_t d = t4 - (e & 0x7FFu);
                    adr[i] = min(d, d + (uint32_t)LAT_ROW) + ((e >> 11) & 0x3FFFFu);
                    const bool absent = r == 0 && t == 0 && js == (uint32_t)(DEG - 1);  // p_{c-1} of check 0 does not exist
                    float v = valid ? *lat_f(adr[i]) : INFINITY;
                    const float mag = (idxo == js) ? c1o : c2o;
                    const uint32_t sgn = (pko << ((32u - DEG) + js)) & SB;
                    old[i] = __uint_as_float(__float_as_uint(mag) | sgn);
                    v = v - old[i];
                    if (absent || !valid) v = INFINITY;
                    x[i] = v;
                    const float a = fabsf(v);
                    mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                    mn1 = fminf(mn1, a);
                    sw |= (__float_as_uint(v) >> 31) << ((uint32_t)(DEG - 1) - js);
                }
                // ---- the two halves merge inside their lane pair
                const float o1 = lat_swap(mn1), o2 = lat_swap(mn2);
                const float g1 = fminf(mn1, o1), g2 = fminf(fmaxf(mn1, o1), fminf(mn2, o2));
                const uint32_t sall = sw | lat_swap(sw);
                const uint32_t tot = (uint32_t)(__popc(sall) & 1);
                const float cst1 = g2 * p.alpha, cst2 = g1 * p.alpha;
                if (ncf > 0) __syncthreads();                         // every read of the layer precedes its writes
                float delta[LDPC_FAST_MAXC > H ? H : LDPC_FAST_MAXC];
#pragma unroll
                for (int i = 0; i < H; i++) {
                    const bool has2 = H + i < DEG;
                    const bool valid = act && (part == 0 || has2);
                    const uint32_t js = (uint32_t)(part ? H + i : i);
                    const bool ismin = fabsf(x[i]) == g1;
                    const float mag = ismin ? cst1 : cst2;
                    const uint32_t s = ((tot << 31) ^ __float_as_uint(x[i])) & SB;
                    const float nw = __uint_as_float(__float_as_uint(mag) | s);
                    if (ismin && valid) li = (int)js;
                    const bool absent = r == 0 && t == 0 && js == (uint32_t)(DEG - 1);
                    if (valid && !absent && ((prim >> js) & 1u)) *lat_f(adr[i]) = x[i] + nw;
                    if (i < (int)(sizeof(delta) / sizeof(delta[0]))) delta[i] = nw - old[i];
                }
                // position of the minimum for the packed state: the LAST slot (in slot order) that holds min1
                const int lo = (int)lat_swap((uint32_t)li);
                const int idxn = part ? (li >= 0 ? li : lo) : (lo >= 0 ? lo : li);
                if (act && part == 0) {
                    *lat_f(sa) = cst1; *lat_f(sa + 4u) = cst2;
                    *lat_u(sa + 8u) = (sall ^ (tot ? ((1u << DEG) - 1u) : 0u)) | ((uint32_t)(idxn < 0 ? 0 : idxn) << 27);
                }
                // ---- duplicate edges (conflict entry i is slot i, first half): ordered delta updates, level by level
                if (ncf > 0) {
                    uint32_t prev = 0u;
                    for (uint32_t i = 0; i < ncf; i++) {
                        const uint32_t lvl = T[48 + i] >> 8, e = T[32 + i];
                        if (lvl != prev) { __syncthreads(); prev = lvl; }
                        if (act && part == 0) {
                            const uint32_t d = t4 - (e & 0x7FFu), a = min(d, d + (uint32_t)LAT_ROW) + ((e >> 11) & 0x3FFFFu);
                            float dl = 0.f;
#pragma unroll
                            for (int k = 0; k < (int)(sizeof(delta) / sizeof(delta[0])); k++) if ((uint32_t)k == i) dl = delta[k];
                            *lat_f(a) = *lat_f(a) + dl;
                        }
                    }
                }
                __syncthreads();                                      // end of the layer
            }
            it++;
            if (p.early_stop || it == p.n_ite) {
                // ---- syndrome of the hard decisions: every check, both halves
                uint32_t bad = 0u;
                for (int r = 0; r < q; r++) {
                    const lat_const_u32 T = tab + r * LDPC_FAST_STRIDE;
                    uint32_t xs = 0u;
#pragma unroll
                    for (int i = 0; i < H; i++) {
                        const bool has2 = H + i < DEG;
                        const uint32_t e = part ? (has2 ? T[H + i < DEG ? H + i : 0] : 0u) : T[i];
                        const bool valid = act && (part == 0 || has2);
                        const uint32_t js = (uint32_t)(part ? H + i : i);
                        const uint32_t d = t4 - (e & 0x7FFu), a = min(d, d + (uint32_t)LAT_ROW) + ((e >> 11) & 0x3FFFFu);
                        const bool absent = r == 0 && t == 0 && js == (uint32_t)(DEG - 1);
                        if (valid && !absent) xs ^= (*lat_f(a) < 0.f) ? SB : 0u;          // (the oracle's hard decision: L < 0, so -0 counts as 0)
                    }
                    xs ^= lat_swap(xs);
                    bad |= xs >> 31;
                }
                if (__any(bad != 0u) && (L & 63) == 0) *lat_u(flag_a) = 1u;
                __syncthreads();
                ok = *lat_u(flag_a) == 0u;
                __syncthreads();
                if (L == 0) *lat_u(flag_a) = 0u;
                if (ok) break;
            }
        }
        __syncthreads();
        // ---- outputs: hard decisions of the K systematic bits (int32 socket and / or packed words), posteriors in natural order, CWD, iteration count
        if (p.bits) { int32_t *V = p.bits + (size_t)f * K; for (int i = L; i < K; i += LAT_THREADS) __builtin_nontemporal_store((int32_t)(*lat_f((uint32_t)i * 4u) < 0.f ? 1 : 0), V + i); }
        if (p.packed) {
            const int nw_ = (K + 31) / 32;
            uint32_t *W = p.packed + (size_t)f * nw_;
            for (int w = L; w < nw_; w += LAT_THREADS) {
                uint32_t word = 0u;
                for (int b = 0; b < 32; b++) { const int i = 32 * w + b; if (i < K && *lat_f((uint32_t)i * 4u) < 0.f) word |= 1u << b; }
                W[w] = word;
            }
        }
        if (p.post) {
            float *P = p.post + (size_t)f * N;
            for (int i = L; i < N; i += LAT_THREADS) {
                uint32_t a;
                if (i < K) a = (uint32_t)i * 4u;
                else { const int c = i - K, tt = c / q, r = c - tt * q; a = (uint32_t)((n_info + r) * LDPC_Z + tt) * 4u; }
                P[i] = *lat_f(a);
            }
        }
        if (L == 0) { if (p.cwd) p.cwd[f] = ok ? 1 : 0; if (p.ites) p.ites[f] = it; }
        __syncthreads();                                              // the image is reused by the next frame
    }
}

// the LDS bytes the kernel needs for this plan (image + junk + inf rows, packed state, one flag word), or 0 if the plan is not the LDS-only one
size_t ldpc_lat_lds_bytes(const LdpcPlan &pl)
{
    if (!pl.fast || !pl.fast_wg8 || pl.fast_mode != 0 || pl.spa) return 0;
    return (size_t)(pl.n_groups + 2) * LAT_ROW + (size_t)pl.M * 12 + 16;
}

hipError_t ldpc_lat_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s)
{
    p.w8.tab = pl.d_w8_tab;
    p.w8.lds_junk = (uint32_t)(pl.n_groups * LAT_ROW);
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.n_info = pl.n_info; p.n_groups = pl.n_groups;
    const size_t lds = ldpc_lat_lds_bytes(pl);
    if (!lds || (uint32_t)pl.w8_lds_junk != p.w8.lds_junk) return hipErrorInvalidValue;      // (the plan's image has the rows in bit-group order: the tables' offsets assume it)
    const int grid = p.n_frames < pl.n_cus ? p.n_frames : pl.n_cus;
#define LAT_CASE(D) \
    if (pl.fast_deg == D) { \
        static size_t configured[64] = {0}; \
        int dev = 0; (void)hipGetDevice(&dev); \
        if (lds > configured[dev & 63]) { \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_lat_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e; \
            configured[dev & 63] = lds; \
        } \
        hipLaunchKernelGGL(ldpc_lat_kernel<D>, dim3(grid), dim3(LAT_THREADS), lds, s, p); \
        return hipGetLastError(); \
    }
    LAT_CASE(27) LAT_CASE(13) LAT_CASE(11)
#undef LAT_CASE
    return hipErrorInvalidValue;
}

}  // namespace dvbs2
