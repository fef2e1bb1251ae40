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
#include "ldpc_det.h"

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
    float alpha, spa_cap;
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

// ---- the sum-product decoders in the reference's sweep order (round 6): one lane per frame like ldpc_nat_kernel, the c->v messages kept per edge (fp32 rows behind the
// posterior rows: [check][slot][64 frames]; the first iteration reads none).  RULE 2: AFF3CT's tanh-product check node (ldpc_det.h) -- with the slots in address-table
// order (k_ldpc.hip: info edges, p_c, p_{c-1}) every operation is the oracle's decoder (ORC_SPA_TANH under ORC_SCHED_NATURAL) in the oracle's order: BIT-EXACT, and as close
// to what the reference's `Decoder_LDPC_BP_horizontal_layered<.., Update_rule_SPA>` does as this library can get.  RULE 1: exact boxplus (forward / backward recursions on
// the hardware exp2 / log2 units), every message clipped to p.spa_cap (`--dec-implem SPA`: AFF3CT's cap; SPA_EXACT: +inf): within 1e-4 max(1, |L|) of the oracle.
// A validation mode: a frame's checks are a serial chain of memory round trips (~1 us each), so a launch wants tens of thousands of frames.
__device__ __forceinline__ float nat_boxplus(float a, float b)
{
    const float e1 = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(a + b)), e2 = __builtin_amdgcn_exp2f(-1.44269504088896341f * fabsf(a - b));
    const float l = (__builtin_amdgcn_logf(1.0f + e1) - __builtin_amdgcn_logf(1.0f + e2)) * 0.693147180559945309f;
    const float mn = fminf(fabsf(a), fabsf(b));
    const float sg = __uint_as_float((__float_as_uint(mn) & 0x7FFFFFFFu) | ((__float_as_uint(a) ^ __float_as_uint(b)) & 0x80000000u));
    return a == INFINITY ? b : (b == INFINITY ? a : sg + l);
}

template <int DEG, int RULE>
__global__ void __launch_bounds__(64)
ldpc_nat_spa_kernel(const NatParams p)
{
    const int g = blockIdx.x, lane = threadIdx.x, f = g * 64 + lane;
    float *W = p.work + (size_t)g * p.grp_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(W, 0, p.grp_words * 4, 0x00020000);
    const uint32_t vo = (uint32_t)lane * 4u;
    const const_u32 tab = (const_u32)p.tab, haz = (const_u32)p.haz;
    const int q = p.q, M = p.M;
    const uint32_t inf_row = (uint32_t)p.N * NAT_ROW, junk_row = inf_row + NAT_ROW, st0 = junk_row + NAT_ROW;
    auto gld = [&](uint32_t soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, soff, 0)); };
    auto gst = [&](uint32_t soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, vo, soff, 0); };
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
        float fwd = INFINITY;
        for (int c = 0; c < M; c++) {
            const const_u32 T = tab + (size_t)r * DEG * 2;
            if ((haz[c >> 5] >> (c & 31)) & 1u) __builtin_amdgcn_s_waitcnt(0);
            uint32_t row[DEG];
            float x[DEG], old[DEG];
            const uint32_t mrow = st0 + (uint32_t)(c * DEG) * NAT_ROW;
#pragma unroll
            for (int j = 0; j < DEG; j++) row[j] = row_of(T, j, t, j == DEG - 1 && c == 0);
#pragma unroll
            for (int j = 0; j < DEG - 1; j++) x[j] = gld(row[j]);
            x[DEG - 1] = c == 0 ? INFINITY : fwd;
#pragma unroll
            for (int j = 0; j < DEG; j++) old[j] = it > 0 ? gld(mrow + (uint32_t)j * NAT_ROW) : 0.f;
#pragma unroll
            for (int j = 0; j < DEG; j++) x[j] = x[j] - old[j];          // (+inf - 0 for an absent slot)
            float nw[DEG];
            if constexpr (RULE == 2) {
                float tv[DEG], P = 1.0f;
                uint32_t sx = 0u;
#pragma unroll
                for (int j = 0; j < DEG; j++) { tv[j] = w8_det_tanh_half(fabsf(x[j])); sx ^= __float_as_uint(x[j]); P = P * tv[j]; }
#pragma unroll
                for (int j = 0; j < DEG; j++) {
                    float val = P / tv[j];
                    val = (val < 1.0f) ? val : __uint_as_float(0x3F7FFFFEu);
                    const float o = w8_det_log1p((val + val) / (1.0f - val));
                    nw[j] = __uint_as_float((__float_as_uint(o) & 0x7FFFFFFFu) | ((sx ^ __float_as_uint(x[j])) & 0x80000000u));
                }
            } else {
                float fw[DEG], bw;
                fw[0] = INFINITY;
#pragma unroll
                for (int j = 1; j < DEG; j++) fw[j] = nat_boxplus(fw[j - 1], x[j - 1]);
                bw = INFINITY;
#pragma unroll
                for (int j = DEG - 1; j >= 0; j--) {
                    const float o = nat_boxplus(fw[j], bw);
                    nw[j] = __uint_as_float((__float_as_uint(fminf(fabsf(o), p.spa_cap)) & 0x7FFFFFFFu) | (__float_as_uint(o) & 0x80000000u));
                    bw = nat_boxplus(bw, x[j]);
                }
            }
            if (live) {
#pragma unroll
                for (int j = 0; j < DEG; j++) {
                    gst(row[j] == inf_row ? junk_row : row[j], x[j] + nw[j]);
                    gst(mrow + (uint32_t)j * NAT_ROW, row[j] == inf_row ? 0.f : nw[j]);
                }
            }
            fwd = x[DEG - 2] + nw[DEG - 2];              // slot DEG-2 is p_c: the next check's p_{c-1}
            if (++r == q) { r = 0; t++; }
        }
        it++;
        if (live) my_ite = it;
        if (p.early_stop || it == p.n_ite) {
            __builtin_amdgcn_s_waitcnt(0);
            uint32_t bad = 0u;
            int r2 = 0, t2 = 0;
            for (int c = 0; c < M; c++) {
                const const_u32 T = tab + (size_t)r2 * DEG * 2;
                uint32_t xs = 0u;
#pragma unroll
                for (int j = 0; j < DEG; j++) xs ^= __float_as_uint(gld(row_of(T, j, t2, j == DEG - 1 && c == 0)));
                bad |= xs >> 31;
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

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same sweep with a check's edges SPLIT OVER PARTS ADJACENT LANES and G = 64 / PARTS frames per wave -- for batches that do not fill the machine with one
// lane per frame (the BASELINE batch of 4096 normal frames is 64 waves there: 26 k frames/s, below the host CPU of the same bench line).  Lane (fl, p) = frame fl of the
// wave's group, part p: it takes slots p SPP .. p SPP + SPP - 1 of the current check (SPP = ceil(DEG / PARTS); slots beyond DEG are NULL), folds them into {min1, min2,
// parity, its sign bits}, and the PARTS lanes of a frame merge those with DPP exchanges inside their quad / half-row (no LDS, no barrier): the two smallest of the union
// are min(a1, b1) and min(max(a1, b1), a2, b2) -- the same two fp32 values the one-lane scan finds -- so every message, posterior and state word is bit for bit the
// one-lane kernel's (and the oracle's ORC_SCHED_NATURAL).  The chain p_c -> p_{c-1} still travels in a register (handed round the frame's lanes by the same exchanges).
// On top: check c + 1's posteriors and state are REQUESTED BEFORE check c is computed (the sweep is a chain of ~2 us memory round trips otherwise), except where the
// host's second hazard plane says that check c + 1 shares a bit with check c or c - 1.
// Image: [bit][G frames] rows of 4 G bytes, same row order as above.
template <int PARTS> __device__ __forceinline__ float nat_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }     // quad_perm [1,0,3,2]
template <int PARTS> __device__ __forceinline__ float nat_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }     // quad_perm [2,3,0,1]
template <int PARTS> __device__ __forceinline__ float nat_mir8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }    // row_half_mirror: lane i <-> 7 - i

// AHEAD = NAT_AHEAD (8 lanes per frame; half of it with 4): the loads of check c + AHEAD go out when check c has been computed -- only for codes whose second hazard plane is empty (no check shares a bit
// with one of the NAT_HAZ_WINDOW before it: every DVB-S2 code of this library); the loop body is then free of branches and unrolled AHEAD times over a ring of AHEAD requests in registers.  AHEAD = 1: any code -- check c + 1 is requested behind a drain of check c's stores.
template <int DEG, int PARTS, int AHEAD>
__global__ void __launch_bounds__(64)
ldpc_nat_part_kernel(const NatParams p)
{
    constexpr int G = 64 / PARTS, SPP = (DEG + PARTS - 1) / PARTS, DEGP = SPP * PARTS;
    constexpr uint32_t RB = G * 4;                 // bytes per row
    extern __shared__ uint32_t s_tab[];            // [q][DEGP][2]: per slot {byte offset of element 0 of its bit run | NULL: the +inf row ; t0 | byte stride between elements << 16}
    const int g = blockIdx.x, lane = threadIdx.x, part = lane % PARTS, fl = lane / PARTS, f = g * G + fl;
    float *W = p.work + (size_t)g * p.grp_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(W, 0, p.grp_words * 4, 0x00020000);
    const uint32_t vo = (uint32_t)fl * 4u;
    const int q = p.q, M = p.M;
    const uint32_t inf_row = (uint32_t)p.N * RB, junk_row = inf_row + RB, st0 = junk_row + RB;
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));
    for (int i = lane; i < q * DEGP; i += 64) {
        const int r = i / DEGP, j = i - r * DEGP;
        const uint32_t e = j < DEG ? p.tab[(size_t)(r * DEG + j) * 2] : (1u << 17), A = j < DEG ? p.tab[(size_t)(r * DEG + j) * 2 + 1] : 0u;
        const bool null = ((e >> 17) & 1u) != 0u;
        const uint32_t stride = null ? 0u : (((e >> 16) & 1u) ? (uint32_t)q : 1u) * RB;
        s_tab[2 * i] = null ? inf_row : A * RB;
        s_tab[2 * i + 1] = (null ? 0u : (e & 0xFFFFu)) | (stride << 16) | (null ? 0x80000000u : 0u);      // (stride < 2^15: q * RB <= 135 * 256)
    }
    __syncthreads();
    auto gldv = [&](uint32_t voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0)); };
    auto gstv = [&](uint32_t voff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, voff, 0, 0); };
    const int js0 = part * SPP;                    // this lane's first slot
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
    const uint32_t tab_lane = (uint32_t)js0 * 8u;  // byte offset of this lane's first slot inside a layer's table row
    // byte offset (row + frame) of local slot k of check (r, t): base + ((t - t0) mod 360) x stride + 4 fl
    auto addr_of = [&](int r, int t, int k, bool &real) -> uint32_t {
        const u32x2 e = *(lds_u32x2 *)(size_t)((uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)s_tab + (uint32_t)(r * DEGP) * 8u + tab_lane + (uint32_t)k * 8u);
        int elem = t - (int)(e.y & 0xFFFFu);
        elem += (elem >> 31) & LDPC_Z;
        real = (int)e.y >= 0;
        return e.x + __umul24((uint32_t)elem, (e.y >> 16) & 0x7FFFu) + vo;
    };
    auto merge_min = [&](float &m1, float &m2, float o1, float o2) { const float hi = fmaxf(m1, o1); m1 = fminf(m1, o1); m2 = fminf(fminf(m2, o2), hi); };

    bool live = f < p.F, ok = false;
    int it = 0, my_ite = 0;
    struct Req { uint32_t a[SPP]; float v[SPP]; float c1, c2, pk; };
    auto request = [&](Req &R, int c, int r, int t) {       // the loads of check c = (r, t): posteriors of this lane's slots (not the forwarded one), the check's packed state
#pragma unroll
        for (int k = 0; k < SPP; k++) {
            bool real;
            const uint32_t a = addr_of(r, t, k, real);
            const bool fwd_slot = js0 + k == DEG - 1;                      // p_{c-1}: comes from check c - 1 in a register, its row is only stored to
            R.a[k] = (real && !(fwd_slot && c == 0)) ? a : junk_row + vo;  // where pass 2 stores (NULL slots and the absent p_{c-1} of check 0: the junk row)
            R.v[k] = gldv(fwd_slot ? inf_row + vo : a);                    // (a NULL slot's `a` is the +inf row already)
        }
        const uint32_t srow = st0 + (uint32_t)c * 3u * RB + vo;
        R.c1 = gldv(srow); R.c2 = gldv(srow + RB); R.pk = gldv(srow + 2 * RB);
    };
    while (it < p.n_ite) {
        float fwd = INFINITY;                     // posterior of p_{c-1} as check c-1 left it (every lane of the frame holds it)
        __builtin_amdgcn_s_waitcnt(0);
        Req ring[AHEAD];
        int rq = 0, tq = 0, cq = 0;               // the request stream: the next check to ask for (AHEAD checks in front of the one being computed)
        auto request_next = [&](Req &R) {
            request(R, cq, rq, tq);
            if (cq + 1 < M) { cq++; if (++rq == q) { rq = 0; tq++; } }      // beyond the last check: the last one again (harmless; every check issues the same instructions)
        };
#pragma unroll
        for (int u = 0; u < AHEAD; u++) request_next(ring[u]);
        int r = 0, t = 0;
        const bool st_lane = part == 0;
        for (int c0 = 0; c0 < M; c0 += AHEAD) {
#pragma unroll
            for (int u = 0; u < AHEAD; u++) {
                const int c = c0 + u;
                Req &R = ring[u];
                // (The compiler drains the vector memory queue once per trip of the unrolled loop -- at its head it cannot bound what is in flight -- and needs no wait for the
                // other AHEAD - 1 checks of the trip, whose loads went out a whole trip earlier: the memory round trip is paid once per AHEAD checks.)
                // ---- this lane's slots of check c
                float v[SPP];
                const float c1o = R.c1, c2o = R.c2;
                const uint32_t pko = __float_as_uint(R.pk), idxo = pko >> 27;
                float mn1 = INFINITY, mn2 = INFINITY;
                uint32_t sacc = 0u;
#pragma unroll
                for (int k = 0; k < SPP; k++) {
                    const uint32_t js = (uint32_t)(js0 + k);
                    float val = R.v[k];
                    if (js0 + k == DEG - 1) val = c == 0 ? INFINITY : fwd;
                    const float mag = (idxo == js) ? c1o : c2o;
                    const float old = nat_and_or(pko << (((32u - DEG) + js) & 31u), SB, mag);
                    float x = val - old;
                    if (js0 + k >= DEG) x = INFINITY;                          // NULL slot of the padding
                    v[k] = x;
                    const float a = fabsf(x);
                    mn2 = __builtin_amdgcn_fmed3f(mn1, mn2, a);
                    mn1 = fminf(mn1, a);
                    sacc = __builtin_amdgcn_alignbit(sacc, __float_as_uint(x), 31);
                }
                // ---- merge over the PARTS lanes of the frame: minima, parity, the sign word
                uint32_t sw = sacc << (uint32_t)(DEGP - js0 - SPP);             // this part's sign bits where slot js has bit DEGP - 1 - js
                merge_min(mn1, mn2, nat_xor1<PARTS>(mn1), nat_xor1<PARTS>(mn2));
                sw |= __float_as_uint(nat_xor1<PARTS>(__uint_as_float(sw)));
                merge_min(mn1, mn2, nat_xor2<PARTS>(mn1), nat_xor2<PARTS>(mn2));
                sw |= __float_as_uint(nat_xor2<PARTS>(__uint_as_float(sw)));
                if (PARTS == 8) {
                    merge_min(mn1, mn2, nat_mir8<PARTS>(mn1), nat_mir8<PARTS>(mn2));
                    sw |= __float_as_uint(nat_mir8<PARTS>(__uint_as_float(sw)));
                }
                sw >>= (uint32_t)(DEGP - DEG);                                  // slot js at bit DEG - 1 - js, as the one-lane kernel packs it (the padding's zero bits fall off)
                const float cst1 = mn2 * p.alpha, cst2 = mn1 * p.alpha;
                const uint32_t tot = (uint32_t)(__popc(sw) & 1);
                uint32_t pkn = sw ^ (tot ? ((1u << DEG) - 1u) : 0u);
                int idxn = 0;                                                    // the LAST slot that holds the minimum (the one-lane scan's rule); none in this part: 0, as there
                float m1s = __uint_as_float(__float_as_uint(cst1) | (tot << 31)), m2s = __uint_as_float(__float_as_uint(cst2) | (tot << 31));
                asm volatile("" : "+v"(m1s), "+v"(m2s));
#pragma unroll
                for (int k = 0; k < SPP; k++) {
                    const float x = v[k];
                    const bool ismin = fabsf(x) == mn1;
                    const float mag = ismin ? m1s : m2s;
                    const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & SB));
                    idxn = ismin ? js0 + k : idxn;
                    v[k] = x + nw;
                }
                {
                    int o = __builtin_amdgcn_update_dpp(0, idxn, 0xB1, 0xF, 0xF, true); idxn = idxn > o ? idxn : o;
                    o = __builtin_amdgcn_update_dpp(0, idxn, 0x4E, 0xF, 0xF, true); idxn = idxn > o ? idxn : o;
                    if (PARTS == 8) { o = __builtin_amdgcn_update_dpp(0, idxn, 0x141, 0xF, 0xF, true); idxn = idxn > o ? idxn : o; }
                }
                pkn |= (uint32_t)idxn << 27;
                // stores: always issued (a frame that has converged, or lies beyond the batch, writes the junk row), the state by the frame's first lane
#pragma unroll
                for (int k = 0; k < SPP; k++) gstv(live ? R.a[k] : junk_row + vo, v[k]);
                {
                    const uint32_t srow = (live && st_lane) ? st0 + (uint32_t)c * 3u * RB + vo : junk_row + vo;
                    const uint32_t sr1 = (live && st_lane) ? RB : 0u;
                    gstv(srow, cst1); gstv(srow + sr1, cst2); gstv(srow + 2 * sr1, __uint_as_float(pkn));
                }
                {   // p_c's new posterior is the next check's p_{c-1}: handed to every lane of the frame (whichever part holds slot DEG-1 picks it up) as an OR over the parts
                    uint32_t fb = 0u;
#pragma unroll
                    for (int k = 0; k < SPP; k++) if (js0 + k == DEG - 2) fb = __float_as_uint(v[k]);
                    fb |= __float_as_uint(nat_xor1<PARTS>(__uint_as_float(fb)));
                    fb |= __float_as_uint(nat_xor2<PARTS>(__uint_as_float(fb)));
                    if (PARTS == 8) fb |= __float_as_uint(nat_mir8<PARTS>(__uint_as_float(fb)));
                    fwd = __uint_as_float(fb);
                }
                // the slot is free: check c + AHEAD goes out now (beyond the last check: check M - 1 again, harmless, so that every check issues the same instructions)
                if (AHEAD == 1) __builtin_amdgcn_s_waitcnt(0);                  // any code: the stores first
                request_next(R);
                if (++r == q) { r = 0; t++; }
            }
        }
        it++;
        if (live) my_ite = it;
        if (p.early_stop || it == p.n_ite) {
            // ---- syndrome of the hard decisions, every check of the frame (the parity of a check is the XOR over its parts: combined per check, before the OR over the checks)
            __builtin_amdgcn_s_waitcnt(0);
            uint32_t bad = 0u;
            int r2 = 0, t2 = 0;
            for (int c = 0; c < M; c++) {
                uint32_t x = 0u;
#pragma unroll
                for (int k = 0; k < SPP; k++) {
                    bool real;
                    const uint32_t a = addr_of(r2, t2, k, real);
                    const float Lv = gldv((js0 + k == DEG - 1 && c == 0) ? inf_row + vo : a);       // (+inf: sign bit 0)
                    x ^= __float_as_uint(Lv);
                }
                x ^= __float_as_uint(nat_xor1<PARTS>(__uint_as_float(x)));
                x ^= __float_as_uint(nat_xor2<PARTS>(__uint_as_float(x)));
                if (PARTS == 8) x ^= __float_as_uint(nat_mir8<PARTS>(__uint_as_float(x)));
                bad |= x >> 31;
                if (++r2 == q) { r2 = 0; t2++; }
            }
            if (live) { ok = bad == 0u; if (ok) live = false; }
            if (!__any(live)) break;
        }
    }
    if (f < p.F && part == 0) {
        if (p.cwd) p.cwd[f] = ok ? 1 : 0;
        if (p.ites) p.ites[f] = my_ite;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------------------
// Round 4, second form for small batches: EIGHT CONSECUTIVE CHECKS of a frame in eight adjacent lanes (lane = 8 x frame-of-the-wave + k, check c0 + k), eight frames per wave --
// the same [bit][8 frames] image as the 8-lanes-per-check kernel above.  A check has 27 edges of which ONE, the parity bit it shares with the check before it, makes the
// sweep serial; the other 26 belong to this check alone among its 25 predecessors (the host's second hazard plane is empty for every code of the library).  So:
//   A. every lane forms v -> c = L - old for its own check's 26 other edges and their two smallest magnitudes, parity and sign word -- 8 checks x 8 frames at once, no exchange;
//   B. the chain runs as a scan over the eight lanes of a frame: in step s every lane takes the posterior of p_{c-1} from its left neighbour (DPP row_shr:1; lane k = 0 from the
//      previous group's last lane), folds that edge in and forms p_c's new posterior; after step s lanes 0 .. s hold their final values (a lane whose input was final repeats
//      the same result), so eight unconditional steps of ~16 instructions leave every lane right -- no select, no LDS, no barrier;
//   C. every lane forms the new messages and posteriors of its check and stores them.
// Per frame and check the wave issues ~11 instructions instead of the ~26 of the form above (no merge rounds, no padding slots, every lane busy), and the loads of a group of
// eight checks go out NAT_CK_AHEAD groups ahead (the last check of the group in flight is 8 NAT_CK_AHEAD + 7 <= NAT_HAZ_WINDOW checks in front of the first one not yet written).
// Same fp32 operations per edge, the same two minima (the two smallest of a set do not depend on the order they are found in), the index of the LAST slot that holds the
// minimum: bit for bit the one-lane kernel and the oracle's ORC_SCHED_NATURAL.
#ifndef NAT_CK_AHEAD_N
#define NAT_CK_AHEAD_N 2
#endif
constexpr int NAT_CK_AHEAD = NAT_CK_AHEAD_N;
#ifndef NAT_CK_TRIP
#define NAT_CK_TRIP 12
#endif
static_assert(NAT_CK_TRIP % NAT_CK_AHEAD == 0, "the ring has to be back at its start when a trip ends");
static_assert(8 * NAT_CK_AHEAD + 7 <= NAT_HAZ_WINDOW, "the requests run further ahead than the host's hazard plane covers");
// WV waves per workgroup share an image whose rows hold the posteriors of 64 / CK x WV frames (a 128-byte line for CK = 8, WV = 4): every wave takes its own 64 / CK frames
// through the same checks at about the same time, so a line is fetched over the fabric once and found in the CU's L1 / the XCD's L2 by the other waves -- with rows of 32
// bytes (one wave's eight frames) the form was bound by the 128 bytes the fabric moves per row touched (1.8 TB/s of its own bytes at every batch size).
template <int DEG, int CK, int WV>
__global__ void __launch_bounds__(64 * WV)
ldpc_nat_ck_kernel(const NatParams p)
{
    constexpr int GW = 64 / CK, G = GW * WV, AH = NAT_CK_AHEAD;      // frames per wave, frames per workgroup = per image row
    static_assert(CK == 8 || CK == 4, "checks side by side");
    constexpr uint32_t RB = G * 4;                 // bytes per row
    extern __shared__ uint32_t s_tab[];            // [q][DEG][2]: per slot {byte offset of element 0 of its bit run | NULL: the +inf row ; t0 | byte stride between elements << 16 | NULL << 31}
    const int g = blockIdx.x, lane = threadIdx.x & 63, k = lane & (CK - 1), fl = lane / CK + GW * (int)(threadIdx.x >> 6), f = g * G + fl;
    float *W = p.work + (size_t)g * p.grp_words;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(W, 0, p.grp_words * 4, 0x00020000);
    const uint32_t vo = (uint32_t)fl * 4u;
    const int q = p.q, M = p.M;
    const uint32_t inf_row = (uint32_t)p.N * RB, junk_row = inf_row + RB, st0 = junk_row + RB;
    uint32_t SB = 0x80000000u;
    asm volatile("" : "+s"(SB));
    for (int i = threadIdx.x; i < q * DEG; i += 64 * WV) {
        const uint32_t e = p.tab[(size_t)i * 2], A = p.tab[(size_t)i * 2 + 1];
        const bool null = ((e >> 17) & 1u) != 0u;
        const uint32_t stride = null ? 0u : (((e >> 16) & 1u) ? (uint32_t)q : 1u) * RB;
        s_tab[2 * i] = null ? inf_row : A * RB;
        s_tab[2 * i + 1] = (null ? 0u : (e & 0xFFFFu)) | (stride << 16) | (null ? 0x80000000u : 0u);      // (stride < 2^15: q * RB <= 135 * 128)
    }
    __syncthreads();
    auto gldv = [&](uint32_t voff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0)); };
    auto gstv = [&](uint32_t voff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rs, voff, 0, 0); };
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
    const uint32_t tab_base = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)s_tab;
    auto addr_of = [&](int r, int t, int j, bool &real) -> uint32_t {
        const u32x2 e = *(lds_u32x2 *)(size_t)(tab_base + (uint32_t)(r * DEG + j) * 8u);
        int elem = t - (int)(e.y & 0xFFFFu);
        elem += (elem >> 31) & LDPC_Z;
        real = (int)e.y >= 0;
        return e.x + __umul24((uint32_t)elem, (e.y >> 16) & 0x7FFFu) + vo;
    };
    const int dr = CK % q, dt = CK / q;            // (r, t) of check c + 8 from those of check c
    bool live = f < p.F, ok = false;
    int it = 0, my_ite = 0;
    struct Req { uint32_t a[DEG]; float v[DEG - 1]; float c1, c2, pk; };
    auto request = [&](Req &R, int c, int r, int t) {       // this lane's check c = (r, t): the posteriors of its slots but the forwarded one, its packed state, and where pass C stores
#pragma unroll
        for (int j = 0; j < DEG; j++) {
            bool real;
            const uint32_t a = addr_of(r, t, j, real);
            R.a[j] = (real && !(j == DEG - 1 && c == 0)) ? a : junk_row + vo;      // (NULL slots and the absent p_{c-1} of check 0: the junk row)
            if (j < DEG - 1) R.v[j] = gldv(a);                                     // (a NULL slot's `a` is the +inf row already)
        }
        const uint32_t srow = st0 + (uint32_t)c * 3u * RB + vo;
        R.c1 = gldv(srow); R.c2 = gldv(srow + RB); R.pk = gldv(srow + 2 * RB);
    };
    const int ngrp = M / CK;
    while (it < p.n_ite) {
        float carry = INFINITY;                   // posterior of p_{c0 - 1} as the previous group's last check left it (check 0: absent, +inf)
        __builtin_amdgcn_s_waitcnt(0);
        Req ring[AH];
        int cq = k, rq = k % q, tq = k / q;        // the request stream of this lane: its check of the next group to ask for
        auto request_next = [&](Req &R) {
            request(R, cq, rq, tq);
            if (cq + CK < M) { cq += CK; rq += dr; tq += dt; if (rq >= q) { rq -= q; tq++; } }      // beyond the last group: the last one again (harmless)
        };
#pragma unroll
        for (int u = 0; u < AH; u++) request_next(ring[u]);
        // (the compiler drains the vector memory queue at the head of the loop -- it cannot bound what is in flight there: a whole memory round trip per trip -- so a trip is
        // NAT_CK_TRIP groups long, the ring of AH requests going round inside it; 12 against 6: 8192 normal frames 41.7 -> 37.2 ms, 4096 frames 37.7 -> 36.8)
        for (int g0 = 0; g0 < ngrp; g0 += NAT_CK_TRIP) {
            if (WV > 1) __builtin_amdgcn_s_barrier();      // the waves of a workgroup stay within a trip of each other: a line one of them has fetched is still near when the others ask
#pragma unroll
            for (int u = 0; u < NAT_CK_TRIP; u++) {
                if (g0 + u >= ngrp) break;
                const int c = (g0 + u) * CK + k;
                Req &R = ring[u % AH];
                // ---- A. the 26 edges that are this check's own
                float v[DEG];
                const float c1o = R.c1, c2o = R.c2;
                const uint32_t pko = __float_as_uint(R.pk), idxo = pko >> 27;
                float pm1 = INFINITY, pm2 = INFINITY;
                uint32_t psacc = 0u;
#pragma unroll
                for (int j = 0; j < DEG - 1; j++) {
                    const float mag = (idxo == (uint32_t)j) ? c1o : c2o;
                    const float old = nat_and_or(pko << ((32u - DEG) + j), SB, mag);
                    const float x = R.v[j] - old;
                    v[j] = x;
                    const float a = fabsf(x);
                    pm2 = __builtin_amdgcn_fmed3f(pm1, pm2, a);
                    pm1 = fminf(pm1, a);
                    psacc = __builtin_amdgcn_alignbit(psacc, __float_as_uint(x), 31);
                }
                const float oldp = nat_and_or(pko << ((32u - DEG) + (DEG - 1)), SB, (idxo == (uint32_t)(DEG - 1)) ? c1o : c2o);
                const float xc = v[DEG - 2];      // p_c's edge
                // ---- B. the chain over the eight checks of the group
                float lout = 0.f, xl = 0.f, mn1 = 0.f, m1s = 0.f, m2s = 0.f, cst1 = 0.f, cst2 = 0.f;
                uint32_t sacc = 0u, tot = 0u;
#pragma unroll
                for (int s8 = 0; s8 < CK; s8++) {
                    const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lout), 0x111, 0xF, 0xF, false));      // row_shr:1
                    const float lp = k == 0 ? carry : left;
                    xl = lp - oldp;
                    const float a = fabsf(xl);
                    const float mn2 = __builtin_amdgcn_fmed3f(pm1, pm2, a);
                    mn1 = fminf(pm1, a);
                    sacc = __builtin_amdgcn_alignbit(psacc, __float_as_uint(xl), 31);
                    cst1 = mn2 * p.alpha; cst2 = mn1 * p.alpha;
                    tot = (uint32_t)(__popc(sacc) & 1);
                    m1s = __uint_as_float(__float_as_uint(cst1) | (tot << 31)); m2s = __uint_as_float(__float_as_uint(cst2) | (tot << 31));
                    const float mag = (fabsf(xc) == mn1) ? m1s : m2s;
                    lout = xc + __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(xc) & SB));
                }
                // the next group's first lane continues from this group's last one
                carry = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane & ~(CK - 1)) | (CK - 1)) * 4, __builtin_bit_cast(int, lout)));
                // ---- C. new messages and posteriors of this lane's check
                v[DEG - 1] = xl;
                uint32_t pkn = sacc ^ (tot ? ((1u << DEG) - 1u) : 0u), idxn = 0u;
                asm volatile("" : "+v"(m1s), "+v"(m2s));
#pragma unroll
                for (int j = 0; j < DEG; j++) {
                    const float x = v[j];
                    const bool ismin = fabsf(x) == mn1;
                    const float mag = ismin ? m1s : m2s;
                    const float nw = __uint_as_float(__float_as_uint(mag) ^ (__float_as_uint(x) & SB));
                    idxn = ismin ? (uint32_t)j : idxn;
                    gstv(live ? R.a[j] : junk_row + vo, x + nw);      // (slot DEG-2, p_c, leaves before the next lane's slot DEG-1 of the same row: the later check's value stays)
                }
                pkn |= idxn << 27;
                {
                    const uint32_t srow = live ? st0 + (uint32_t)c * 3u * RB + vo : junk_row + vo;
                    const uint32_t sr1 = live ? RB : 0u;
                    gstv(srow, cst1); gstv(srow + sr1, cst2); gstv(srow + 2 * sr1, __uint_as_float(pkn));
                }
                request_next(R);
            }
        }
        it++;
        if (live) my_ite = it;
        if (p.early_stop || it == p.n_ite) {
            // ---- syndrome of the hard decisions: lane k takes checks k, k + 8, ..; the frame's eight lanes OR their findings
            __builtin_amdgcn_s_waitcnt(0);
            uint32_t bad = 0u;
            int r2 = k % q, t2 = k / q;
            for (int c = k; c < M; c += CK) {
                uint32_t x = 0u;
#pragma unroll
                for (int j = 0; j < DEG; j++) {
                    bool real;
                    const uint32_t a = addr_of(r2, t2, j, real);
                    x ^= __float_as_uint(gldv((j == DEG - 1 && c == 0) ? inf_row + vo : a));       // (+inf: sign bit 0)
                }
                bad |= x >> 31;
                r2 += dr; t2 += dt; if (r2 >= q) { r2 -= q; t2++; }
            }
            bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0xB1, 0xF, 0xF, true);
            bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0x4E, 0xF, 0xF, true);
            if (CK == 8) bad |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bad, 0x141, 0xF, 0xF, true);
            if (live) { ok = bad == 0u; if (ok) live = false; }
            if (WV > 1) { if (!__syncthreads_or(live ? 1 : 0)) break; }      // (the waves of a workgroup leave together: they meet at the barriers above)
            else if (!__any(live)) break;
        }
    }
    if (f < p.F && k == 0) {
        if (p.cwd) p.cwd[f] = ok ? 1 : 0;
        if (p.ites) p.ites[f] = my_ite;
    }
}

// ---- llr [F][N] <-> rows of G frames (G = 64 / PARTS), the general forms of nat_load_kernel / nat_store_kernel
__global__ void __launch_bounds__(256)
nat_load_g_kernel(const NatParams p, int G)
{
    __shared__ float tile[64][65];
    const int g = blockIdx.y, n0 = blockIdx.x * 64;
    float *W = p.work + (size_t)g * p.grp_words;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int l = ty; l < G; l += 4) {
        const int f = g * G + l, n = n0 + tx;
        tile[l][tx] = (f < p.F && n < p.N) ? __builtin_nontemporal_load(&p.llr[(size_t)f * p.N + n]) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * G; i += 256) {
        const int k = i / G, l = i - k * G, n = n0 + k;
        if (n < p.N) W[(size_t)n * G + l] = tile[l][k];
    }
    const size_t st0 = (size_t)(p.N + 2) * G, st_words = (size_t)3 * p.M * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < st_words; i += (size_t)gridDim.x * 256) W[st0 + i] = 0.f;
    if (blockIdx.x == 0 && (int)threadIdx.x < G) { W[(size_t)p.N * G + threadIdx.x] = INFINITY; W[(size_t)(p.N + 1) * G + threadIdx.x] = 0.f; }
}

__global__ void __launch_bounds__(256)
nat_store_g_kernel(const NatParams p, int G)
{
    __shared__ float tile[64][65];                  // tile[bit][frame]
    const int g = blockIdx.y, n0 = blockIdx.x * 64;
    const float *W = p.work + (size_t)g * p.grp_words;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * G; i += 256) {
        const int k = i / G, l = i - k * G, n = n0 + k;
        tile[k][l] = n < p.N ? W[(size_t)n * G + l] : 0.f;
    }
    __syncthreads();
    for (int l = ty; l < G; l += 4) {
        const int f = g * G + l, n = n0 + tx;
        if (f >= p.F || n >= p.N) continue;
        const float Lv = tile[tx][l];
        if (p.post) p.post[(size_t)f * p.N + n] = Lv;
        if (p.bits && n < p.K) __builtin_nontemporal_store((int32_t)(Lv < 0.f ? 1 : 0), &p.bits[(size_t)f * p.K + n]);
    }
    if (p.packed && n0 < p.K && (int)threadIdx.x < 2 * G) {
        const int l = threadIdx.x >> 1, h = threadIdx.x & 1, f = g * G + l;
        if (f < p.F) {
            uint32_t word = 0u;
            for (int b = 0; b < 32; b++) { const int n = n0 + 32 * h + b; if (n < p.K && tile[32 * h + b][l] < 0.f) word |= 1u << b; }
            const int n_words = (p.K + 31) / 32, wd = (n0 >> 5) + h;
            if (wd < n_words) p.packed[(size_t)f * n_words + wd] = word;
        }
    }
}

template <int DEG, int CK, int WV>
static hipError_t nat_ck_launch(const LdpcPlan &pl, NatParams p, hipStream_t s)
{
    constexpr int G = 64 / CK * WV;
    p.grp_words = (uint32_t)((size_t)(pl.N + 2 + 3 * pl.M) * G);
    const int groups = (p.F + G - 1) / G, tiles = (pl.N + 63) / 64;
    hipLaunchKernelGGL(nat_load_g_kernel, dim3(tiles, groups), dim3(256), 0, s, p, G);
    hipLaunchKernelGGL((ldpc_nat_ck_kernel<DEG, CK, WV>), dim3(groups), dim3(64 * WV), (size_t)pl.q * DEG * 8, s, p);
    hipLaunchKernelGGL(nat_store_g_kernel, dim3(tiles, groups), dim3(256), 0, s, p, G);
    return hipGetLastError();
}

template <int DEG, int PARTS>
static hipError_t nat_part_launch(const LdpcPlan &pl, NatParams p, hipStream_t s)
{
    bool clean = true;      // the second hazard plane is empty: loads may run NAT_AHEAD checks ahead
    for (size_t i = pl.nat_haz.size() / 2; i < pl.nat_haz.size(); i++) clean &= pl.nat_haz[i] == 0u;
    constexpr int G = 64 / PARTS, SPP = (DEG + PARTS - 1) / PARTS;
    p.grp_words = (uint32_t)((size_t)(pl.N + 2 + 3 * pl.M) * G);
    const int groups = (p.F + G - 1) / G, tiles = (pl.N + 63) / 64;
    hipLaunchKernelGGL(nat_load_g_kernel, dim3(tiles, groups), dim3(256), 0, s, p, G);
    constexpr int AH = PARTS == 8 ? NAT_AHEAD : NAT_AHEAD / 2;
    if (clean && pl.M % AH == 0) hipLaunchKernelGGL((ldpc_nat_part_kernel<DEG, PARTS, AH>), dim3(groups), dim3(64), (size_t)pl.q * SPP * PARTS * 8, s, p);
    else hipLaunchKernelGGL((ldpc_nat_part_kernel<DEG, PARTS, 1>), dim3(groups), dim3(64), (size_t)pl.q * SPP * PARTS * 8, s, p);
    hipLaunchKernelGGL(nat_store_g_kernel, dim3(tiles, groups), dim3(256), 0, s, p, G);
    return hipGetLastError();
}

hipError_t ldpc_nat_launch(const LdpcPlan &pl, const LdpcKParams &kp, float *work, hipStream_t s)
{
    NatParams p;
    p.llr = kp.llr; p.work = work; p.tab = pl.d_nat_tab; p.haz = pl.d_nat_haz;
    p.bits = kp.bits; p.packed = kp.packed; p.cwd = kp.cwd; p.post = kp.post; p.ites = kp.ites;
    p.N = pl.N; p.K = pl.K; p.M = pl.M; p.q = pl.q; p.F = kp.n_frames; p.n_ite = kp.n_ite; p.early_stop = kp.early_stop; p.alpha = kp.alpha;
    if (pl.spa) {      // sum-product in the reference's sweep order: one lane per frame (ldpc_nat_spa_kernel)
        p.spa_cap = pl.spa_rule == 3 ? LDPC_SPA_CAP : INFINITY;
        p.grp_words = (uint32_t)ldpc_nat_group_words(pl);
        const int groups = (kp.n_frames + 63) / 64, tiles = (pl.N + 63) / 64;
        hipLaunchKernelGGL(nat_load_kernel, dim3(tiles, groups), dim3(256), 0, s, p);
#define NAT_SPA_CASE(D) \
        if (pl.fast_deg == D) { if (pl.spa_rule == 2) hipLaunchKernelGGL((ldpc_nat_spa_kernel<D, 2>), dim3(groups), dim3(64), 0, s, p); else hipLaunchKernelGGL((ldpc_nat_spa_kernel<D, 1>), dim3(groups), dim3(64), 0, s, p); }
        NAT_SPA_CASE(27) NAT_SPA_CASE(13) NAT_SPA_CASE(11)
#undef NAT_SPA_CASE
        hipLaunchKernelGGL(nat_store_kernel, dim3(tiles, groups), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    // Form by the size of the batch.  Codes whose second hazard plane is empty and whose check count divides by eight (every DVB-S2 code of the library) run CONSECUTIVE CHECKS
    // side by side (ldpc_nat_ck_kernel).  (round 5) While the batch gives every CU at most ONE workgroup of 16 frames (4096 frames on 256 CUs: the BASELINE batch): 8 checks x 8
    // frames per wave, TWO waves per workgroup -- the sweep's step time follows the number of cache lines the waves of a CU touch per group of checks (every slot of every check
    // is one line access per wave whatever the row holds), and rows of 32 frames (the two forms below) put four waves on half the CUs at this size: 4096 normal frames 35.3 -> 24.9-26.8
    // ms = 115 -> 153-165 k frames/s, 1024 frames 31.4 -> 19.3 ms; with two such workgroups on a CU the time doubles (6144 frames: 47.5 against 36.5 ms), so from there on
    // 4 checks x 16 frames, two waves per workgroup = rows of 32 frames (8192 frames: 217 k; 16384+: 245-260 k; one lane per frame reached 171 k at 32768).  Any other code: one lane per
    // frame when that alone gives every SIMD a wave, else a check's edges over 4 or 8 lanes.  DVBS2HIP_NAT_PARTS = 1 | 4 | 8 (those forms), 88 | 44 | 82 | 81 | 41 | 48 (side by side) overrides.
    bool clean = pl.M % 8 == 0;
    for (size_t i = pl.nat_haz.size() / 2; i < pl.nat_haz.size(); i++) clean &= pl.nat_haz[i] == 0u;
    int parts = clean ? (kp.n_frames <= 8 * pl.n_cus ? 81 : kp.n_frames <= 16 * pl.n_cus ? 82 : kp.n_frames >= 3072 ? 44 : 88)      // (one wave per CU while that holds the batch: 2048 normal frames 20.9 -> 19.4 ms)
                     : kp.n_frames >= 32768 ? 1 : kp.n_frames > 6144 ? 4 : 8;
    if (const char *ev = getenv("DVBS2HIP_NAT_PARTS")) { const int v = atoi(ev); if (v == 1 || v == 4 || v == 8 || ((v == 88 || v == 44 || v == 82 || v == 81 || v == 48 || v == 41) && clean)) parts = v; }
    // (two digits: checks side by side, waves per workgroup -- 88 and 44 are the forms above, whose rows hold 32 frames; 82 = <8, 2> and 41 = <4, 1>: rows of 16 frames, twice the workgroups; 81 = <8, 1>: rows of 8; 48 = <4, 4>: rows of 64)
#define NAT_CK_CASE(code, CKv, WVv) \
    if (parts == code) { \
        if (pl.fast_deg == 27) return nat_ck_launch<27, CKv, WVv>(pl, p, s); \
        if (pl.fast_deg == 13) return nat_ck_launch<13, CKv, WVv>(pl, p, s); \
        return nat_ck_launch<11, CKv, WVv>(pl, p, s); \
    }
    NAT_CK_CASE(88, 8, 4) NAT_CK_CASE(44, 4, 2) NAT_CK_CASE(82, 8, 2) NAT_CK_CASE(81, 8, 1) NAT_CK_CASE(48, 4, 4) NAT_CK_CASE(41, 4, 1)
#undef NAT_CK_CASE
    if (parts > 1) {
        if (pl.fast_deg == 27) return parts == 4 ? nat_part_launch<27, 4>(pl, p, s) : nat_part_launch<27, 8>(pl, p, s);
        if (pl.fast_deg == 13) return parts == 4 ? nat_part_launch<13, 4>(pl, p, s) : nat_part_launch<13, 8>(pl, p, s);
        return parts == 4 ? nat_part_launch<11, 4>(pl, p, s) : nat_part_launch<11, 8>(pl, p, s);
    }
    p.grp_words = (uint32_t)ldpc_nat_group_words(pl);
    const int groups = (kp.n_frames + 63) / 64, tiles = (pl.N + 63) / 64;
    hipLaunchKernelGGL(nat_load_kernel, dim3(tiles, groups), dim3(256), 0, s, p);
    if (pl.fast_deg == 27) hipLaunchKernelGGL(ldpc_nat_kernel<27>, dim3(groups), dim3(64), 0, s, p);
    else if (pl.fast_deg == 13) hipLaunchKernelGGL(ldpc_nat_kernel<13>, dim3(groups), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(ldpc_nat_kernel<11>, dim3(groups), dim3(64), 0, s, p);
    hipLaunchKernelGGL(nat_store_kernel, dim3(tiles, groups), dim3(256), 0, s, p);
    return hipGetLastError();
}

size_t ldpc_nat_group_words(const LdpcPlan &pl) { return (size_t)(pl.N + 2 + (pl.spa ? pl.fast_deg : 3) * pl.M) * 64; }      // (sum-product: one fp32 message per edge slot)

}  // namespace dvbs2
