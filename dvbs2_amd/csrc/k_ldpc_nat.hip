// a1, the reference's own sweep order on the device -- horizontal-layered NMS with the checks of a frame
// visited in NATURAL row order (what AFF3CT's BP_HORIZONTAL_LAYERED does; the oracle's
// ORC_SCHED_NATURAL), bit-exact with it.
//
// Check c reads the posterior that check c-1 has just written (they share parity bit c-1), so a frame is a
// serial chain of M check updates per iteration: there is no parallelism inside a frame to give to lanes.
// This kernel therefore maps the reference's `--dec-simd INTER` idea onto the GPU: ONE LANE PER FRAME, 64
// frames per wavefront, every lane running the same check at the same time.  The image is frame-interleaved
// -- row b = the posteriors of bit b of the wave's 64 frames, 256 contiguous bytes -- so every access of a
// wave is one fully coalesced row, the row address is wave-uniform (SGPR soffset from the layer table, no
// per-lane address arithmetic at all) and the 288 GB of HBM hold the 22 MB a wave needs for N = 64800 many
// times over.  The serial chain p_{c-1} -> check c is forwarded in a register; the few other bits that two
// consecutive checks share are flagged by the host and fenced.
//
// It is a validation and large-batch mode, selected with dvbs2hip_set_ldpc_schedule(.., NATURAL): the data
// path (17 MB of state traffic per frame and decode, no reuse on chip) is HBM-bound and a 4096-frame batch
// fills only 64 of the chip's 1024 SIMDs.  The QC-layer kernels are the throughput path.
#include "dvbs2hip_internal.h"

namespace dvbs2 {

typedef const __attribute__((address_space(4))) uint32_t *const_u32;
constexpr int NAT_ROW = 64 * 4;          // bytes per row: one fp32 per frame of the wave

__device__ __forceinline__ float nat_and_or(uint32_t a, uint32_t m_sgpr, float b)
{
    float r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m_sgpr), "v"(b));
    return r;
}

struct NatParams {
    const float *llr;          // [F][N]
    float *work;               // [groups][rows][64]: N posterior rows | +inf row | junk row | 3 M state rows (c1, c2, pk per check)
    const uint32_t *tab;       // [q][DEG][2]: t0 | stride-q flag << 16 | NULL flag << 17 ; A (bit = A + elem * (flag ? q : 1))
    const uint32_t *haz;       // [ceil(M/32)] bit c: check c shares a bit other than p_{c-1} with the check before it
    int32_t *bits; uint32_t *packed; int8_t *cwd; float *post; int32_t *ites;
    int32_t N, K, M, q, F, n_ite, early_stop;
    float alpha;
    uint32_t grp_words;        // words per group of 64 frames in `work`
};

// ---- llr [F][N] -> frame-interleaved rows; state := 0; +inf row
__global__ void __launch_bounds__(256)
nat_load_kernel(const NatParams p)
{
    __shared__ float tile[64][65];
    const int g = blockIdx.y, n0 = blockIdx.x * 64;
    float *W = p.work + (size_t)g * p.grp_words;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int l = ty; l < 64; l += 4) {
        const int f = g * 64 + l, n = n0 + tx;
        tile[l][tx] = (f < p.F && n < p.N) ? __builtin_nontemporal_load(&p.llr[(size_t)f * p.N + n]) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int n = n0 + k;
        if (n < p.N) W[(size_t)n * 64 + tx] = tile[tx][k];
    }
    // state rows and the two special rows, spread over the blocks of the group
    const size_t st0 = (size_t)(p.N + 2) * 64, st_words = (size_t)3 * p.M * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < st_words; i += (size_t)gridDim.x * 256) W[st0 + i] = 0.f;
    if (blockIdx.x == 0 && threadIdx.x < 64) { W[(size_t)p.N * 64 + threadIdx.x] = INFINITY; W[(size_t)(p.N + 1) * 64 + threadIdx.x] = 0.f; }
}

// ---- the decoder: one wave per group of 64 frames
template <int DEG>
__global__ void __launch_bounds__(64)
ldpc_nat_kernel(const NatParams p)
{
    const int g = blockIdx.x, lane = threadIdx.x, f = g * 64 + lane;
    float *W = p.work + (size_t)g * p.grp_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(W, 0, p.grp_words * 4, 0x00020000);
    const uint32_t vo = (uint32_t)lane * 4u;
    const const_u32 tab = (const_u32)p.tab, haz = (const_u32)p.haz;
    const int q = p.q, M = p.M;
    const uint32_t inf_row = (uint32_t)p.N * NAT_ROW, junk_row = inf_row + NAT_ROW, st0 = junk_row + NAT_ROW;
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));
    auto gld = [&](uint32_t soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, soff, 0)); };
    auto gst = [&](uint32_t soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, vo, soff, 0); };
    // row (byte offset) of slot j of check (r, t); NULL slots and the absent p_{c-1} of check 0 read +inf
    auto row_of = [&](const_u32 T, int j, int t, bool absent) -> uint32_t {
        const uint32_t e = T[2 * j], A = T[2 * j + 1];
        int elem = t - (int)(e & 0xFFFFu);
        elem = elem < 0 ? elem + LDPC_Z : elem;
        const uint32_t bit = A + (uint32_t)elem * ((e >> 16) & 1u ? (uint32_t)q : 1u);
        return (((e >> 17) & 1u) || absent) ? inf_row : bit * NAT_ROW;
    };

    bool live = f < p.F, ok = false;
    int it = 0, my_ite = 0;
    while (it < p.n_ite) {
        int r = 0, t = 0;
        float fwd = INFINITY;                     // posterior of p_{c-1} as check c-1 left it
        for (int c = 0; c < M; c++) {
            const const_u32 T = tab + (size_t)r * DEG * 2;
            if ((haz[c >> 5] >> (c & 31)) & 1u) __builtin_amdgcn_s_waitcnt(0);     // a shared bit: the previous check's stores first
            uint32_t row[DEG];
            float v[DEG];
#pragma unroll
            for (int j = 0; j < DEG; j++) row[j] = row_of(T, j, t, j == DEG - 1 && c == 0);
#pragma unroll
            for (int j = 0; j < DEG - 1; j++) v[j] = gld(row[j]);
            v[DEG - 1] = c == 0 ? INFINITY : fwd;
            const uint32_t srow = st0 + (uint32_t)c * 3u * NAT_ROW;
            const float c1o = gld(srow), c2o = gld(srow + NAT_ROW);
            const uint32_t pko = __float_as_uint(gld(srow + 2 * NAT_ROW));
            float mn1 = INFINITY, mn2 = INFINITY;
            uint32_t sacc = 0u;
            const uint32_t idxo = pko >> 27;
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const float mag = (idxo == (uint32_t)j) ? c1o : c2o;
                const float old = nat_and_or(pko << ((32u - DEG) + j), SB, mag);
                const float x = v[j] - old;
                v[j] = x;
                const float a = fabsf(x);
                mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                mn1 = fminf(mn1, a);
                sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(x), 31);
            }
            const float cst1 = mn2 * p.alpha, cst2 = mn1 * p.alpha;
            const uint32_t tot = (uint32_t)(__popc(sacc) & 1);
            uint32_t pkn = sacc ^ (tot ? ((1u << DEG) - 1u) : 0u), idxn = 0u;
            float m1s = __uint_as_float(__float_as_uint(cst1) | (tot << 31)), m2s = __uint_as_float(__float_as_uint(cst2) | (tot << 31));
            asm volatile("" : "+v"(m1s), "+v"(m2s));
#pragma unroll
            for (int j = 0; j < DEG; j++) {
                const float x = v[j];
                const bool ismin = fabsf(x) == mn1;
                const float mag = ismin ? m1s : m2s;
                const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & SB));
                idxn = ismin ? (uint32_t)j : idxn;
                v[j] = x + nw;
            }
            pkn |= idxn << 27;
            if (live) {
#pragma unroll
                for (int j = 0; j < DEG; j++) gst(row[j] == inf_row ? junk_row : row[j], v[j]);
                gst(srow, cst1); gst(srow + NAT_ROW, cst2); gst(srow + 2 * NAT_ROW, __uint_as_float(pkn));
            }
            fwd = v[DEG - 2];                     // slot DEG-2 is p_c: the next check's p_{c-1}
            if (++r == q) { r = 0; t++; }
        }
        it++;
        if (live) my_ite = it;
        if (p.early_stop || it == p.n_ite) {
            // ---- syndrome of the hard decisions, every check of the frame
            __builtin_amdgcn_s_waitcnt(0);
            uint32_t bad = 0u;
            int r2 = 0, t2 = 0;
            for (int c = 0; c < M; c++) {
                const const_u32 T = tab + (size_t)r2 * DEG * 2;
                float Lv[DEG];
#pragma unroll
                for (int j = 0; j < DEG; j++) Lv[j] = gld(row_of(T, j, t2, j == DEG - 1 && c == 0));
                uint32_t x = 0u;
#pragma unroll
                for (int j = 0; j < DEG; j++) x ^= __float_as_uint(Lv[j]);
                bad |= x >> 31;
                if (++r2 == q) { r2 = 0; t2++; }
            }
            if (live) { ok = bad == 0u; if (ok) live = false; }
            if (!__any(live)) break;
        }
    }
    if (f < p.F) {
        if (p.cwd) p.cwd[f] = ok ? 1 : 0;
        if (p.ites) p.ites[f] = my_ite;
    }
}

// ---- frame-interleaved posteriors -> bits [F][K] / packed / post [F][N]
__global__ void __launch_bounds__(256)
nat_store_kernel(const NatParams p)
{
    __shared__ float tile[64][65];
    const int g = blockIdx.y, n0 = blockIdx.x * 64;
    const float *W = p.work + (size_t)g * p.grp_words;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int n = n0 + k;
        tile[k][tx] = n < p.N ? W[(size_t)n * 64 + tx] : 0.f;        // tile[bit][frame]
    }
    __syncthreads();
    for (int l = ty; l < 64; l += 4) {
        const int f = g * 64 + l, n = n0 + tx;
        if (f >= p.F || n >= p.N) continue;
        const float Lv = tile[tx][l];
        if (p.post) p.post[(size_t)f * p.N + n] = Lv;
        if (p.bits && n < p.K) __builtin_nontemporal_store((int32_t)(Lv < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + n]);
    }
    if (p.packed && n0 < p.K && threadIdx.x < 128) {
        // two 32-bit words per frame and tile: bit i of word w = info bit 32 w + i
        const int l = threadIdx.x >> 1, h = threadIdx.x & 1, f = g * 64 + l;
        if (f < p.F) {
            uint32_t word = 0u;
            for (int b = 0; b < 32; b++) { const int n = n0 + 32 * h + b; if (n < p.K && tile[32 * h + b][l] < 0.f) word |= 1u << b; }
            const int n_words = (p.K + 31) / 32, wd = (n0 >> 5) + h;
            if (wd < n_words) p.packed[(size_t)f * n_words + wd] = word;
        }
    }
}

hipError_t ldpc_nat_launch(const LdpcPlan &pl, const LdpcKParams &kp, float *work, hipStream_t s)
{
    NatParams p;
    p.llr = kp.llr; p.work = work; p.tab = pl.d_nat_tab; p.haz = pl.d_nat_haz;
    p.bits = kp.bits; p.packed = kp.packed; p.cwd = kp.cwd; p.post = kp.post; p.ites = kp.ites;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.F = kp.n_frames; p.n_ite = kp.n_ite; p.early_stop = kp.early_stop; p.alpha = kp.alpha;
    p.grp_words = (uint32_t)ldpc_nat_group_words(pl);
    const int groups = (kp.n_frames + 63) / 64, tiles = (pl.N + 63) / 64;
    hipLaunchKernelGGL(nat_load_kernel, dim3(tiles, groups), dim3(256), 0, s, p);
    if (pl.fast_deg == 27) hipLaunchKernelGGL(ldpc_nat_kernel<27>, dim3(groups), dim3(64), 0, s, p);
    else if (pl.fast_deg == 13) hipLaunchKernelGGL(ldpc_nat_kernel<13>, dim3(groups), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(ldpc_nat_kernel<11>, dim3(groups), dim3(64), 0, s, p);
    hipLaunchKernelGGL(nat_store_kernel, dim3(tiles, groups), dim3(256), 0, s, p);
    return hipGetLastError();
}

size_t ldpc_nat_group_words(const LdpcPlan &pl) { return (size_t)(pl.N + 2 + 3 * pl.M) * 64; }

}  // namespace dvbs2
