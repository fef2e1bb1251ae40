// a3 soft demapper, a4 LLR de-interleaver, a6 noise estimator, a7 PL descrambler + header /
// pilot removal, a8 BB descrambler, a9 BER/FER monitor -- the streaming (HBM-bound) kernels of
// the DVB-S2 RX inner path for gfx950.
//
// Reference interfaces replaced (/root/reference):
//   a3  Modem_generic_fast::demodulate            src/common/Factory/DVBS2/DVBS2.cpp:478-488
//   a4  Interleaver<float,uint32_t>::deinterleave src/common/Factory/DVBS2/DVBS2.cpp:451-476
//   a6  Estimator_DVBS2::_estimate                src/common/Module/Estimator/Estimator_DVBS2.hxx:31-58
//   a7  Scrambler_PL::__scramble(false)           src/common/Module/Scrambler/Scrambler_PL/Scrambler_PL.hxx:61-78
//       Framer::_remove_plh                       src/common/Module/Framer/Framer.hxx:330-343
//   a8  Scrambler_BB::_descramble                 src/common/Module/Scrambler/Scrambler_BB/Scrambler_BB.hxx:51-72
//   a9  Monitor_BFER::check_errors                src/common/Factory/DVBS2/DVBS2.cpp:575-591
//
// front_rx_kernel fuses a7 -> a6 -> a3 -> a4: the PL frame is read from HBM once (the second
// sweep hits L2), the index maps of remove_plh and of the column/row de-interleaver are pure
// arithmetic folded into the load and the store, and the only HBM write is the LLR frame the
// LDPC kernel consumes.
#include "dvbs2hip_internal.h"
#include <cstdlib>

namespace dvbs2 {

constexpr int FRONT_THREADS = 256;
constexpr int PL_M = 90, PL_P = 36, PL_SLOTS = 16;

// xfec symbol k -> index of the same symbol inside the PL frame (Framer.hxx:330-343)
__device__ __forceinline__ int pl_index(int k, int n_pilots)
{
    int b = k / (PL_SLOTS * PL_M);
    b = b < n_pilots ? b : n_pilots;
    return PL_M + k + PL_P * b;
}

// multiply by conj(exp(j pi/2 R)) (Scrambler_PL.hxx:66-76 with scr_flag = false)
__device__ __forceinline__ float2 pl_derotate(float2 x, int R)
{
    // R = 0: (x, y)   1: (y, -x)   2: (-x, -y)   3: (-y, x) -- branch-free (a switch becomes four exec-masked paths per wave, every wave holds all four values):
    // odd R swaps the parts, bit 1 of R negates the first output, bit 1 of R + 1 the second
    const bool sw = (R & 1) != 0;
    const float a = sw ? x.y : x.x, b = sw ? x.x : x.y;
    return make_float2(__uint_as_float(__float_as_uint(a) ^ (((uint32_t)R << 30) & 0x80000000u)), __uint_as_float(__float_as_uint(b) ^ (((uint32_t)(R + 1) << 30) & 0x80000000u)));
}

// position of interleaved LLR i = k*bps + b in the natural (code) order
__device__ __forceinline__ int deitl_index(int k, int b, int bps, int cols, int order, int n_rows)
{
    if (cols <= 1) return k * bps + b;
    // every DVB-S2 interleaver has as many columns as the symbol has bits: LLR b of symbol k is row k of column b -- no division
    // (a division by a run-time `cols` is ~25 instructions per LLR, which was a third of the APSK front end's time)
    if (cols == bps) return (order == DVBS2HIP_ITL_TOP_LEFT ? b : cols - 1 - b) * n_rows + k;
    const int i = k * bps + b, row = i / cols, j = i - row * cols;
    return (order == DVBS2HIP_ITL_TOP_LEFT ? j : cols - 1 - j) * n_rows + row;
}

// Exact log-sum-exp demapper:  L_b = max*_{s: bit_b(s)=0}(met_s) - max*_{s: bit_b(s)=1}(met_s),
// met_s = -|y - s|^2 / (2 sigma^2).  The reference (Modem_generic_fast, MAX = max_star) folds
// max*(a,b) = max(a,b) + log1p(exp(-|a-b|)) pairwise; the same quantity is evaluated here with far fewer
// instructions (the kernel is bound by the vector ALU, not by HBM, once a frame has 8+ points per symbol):
//  * |y|^2 is common to all points and drops out of every difference:  met_s = c (2 y.s - |s|^2) + const,
//    c = 1 / (2 sigma^2).  In log2 units u_s = yr A_s + yi B_s + D_s with the per-FRAME table
//    (A, B, D)_s = c log2(e) (2 re s, 2 im s, -|s|^2) in LDS: two FMAs per point, operands broadcast from LDS and
//    shared by the U symbols a lane demaps together;
//  * one exp2 per point relative to the overall maximum, shared by all bits;
//  * the subset sums of all bits from one pairwise tree (bit b splits the level-b partial sums into even / odd):
//    82 additions for 32 points instead of 150;
//  * two log2 and one multiply per bit.
// The exponentials are taken relative to 2^-120 of the overall maximum, so a subset sum is a normal number with full
// relative precision unless the whole subset lies 2^-220 below the maximum (|LLR| beyond ~150: sum < 2^-100); such a
// symbol takes the per-subset form, every term relative to its own subset's maximum.
// |LLR error| vs the pairwise form stays below 1e-4 * max(1, |LLR|) (tests/test_front_gpu.py).
#ifndef FRONT_TAB_SCALAR   // 1: the general demapper takes the frame's constellation table from scalar registers (demap_symbols_s), 0: from LDS (demap_symbols)
#define FRONT_TAB_SCALAR 1
#endif
#ifndef FRONT_BUF_ST
#define FRONT_BUF_ST 1
#endif
#ifndef FRONT_REG_TS       // the register-resident 8PSK kernels: scalar-side table and buffer stores as in front_kernel
#define FRONT_REG_TS 1
#endif
#ifndef FRONT_TAB_VGPR
#define FRONT_TAB_VGPR 0
#endif
#ifndef FRONT_PK           // demap_symbols_s: two points per v_pk_fma_f32 / v_pk_add_f32
#define FRONT_PK 1
#endif
#ifndef FRONT_U_APSK
#define FRONT_U_APSK 1
#endif
#ifndef FRONT_ANALYTIC_REF
#define FRONT_ANALYTIC_REF 1
#endif
constexpr float FRONT_LOG2E = 1.44269504088896341f, FRONT_LN2 = 0.693147180559945309f;

// per-frame table of the constellation, thread s < 2^bps:  tab[s] = c log2(e) (2 re, 2 im, -|s|^2, 0)
__device__ __forceinline__ float4 demap_table_entry(const float *cstl, int s, float inv2s2)
{
    const float k = inv2s2 * FRONT_LOG2E, sr = cstl[2 * s], si = cstl[2 * s + 1];
    return make_float4(2.0f * k * sr, 2.0f * k * si, -k * (sr * sr + si * si), k);      // .w: the scale itself (FRONT_ANALYTIC_REF)
}

template <int BPS>
__device__ __forceinline__ void demap_symbol_far(float yr, float yi, const float4 *tab, float *out)
{
    constexpr int P = 1 << BPS;
    float u[P];
#pragma unroll
    for (int s = 0; s < P; s++) {
        if ((s & 7) == 0) asm volatile("" ::: "memory");          // at most 8 table rows in flight (registers)
        const float4 t = tab[s]; u[s] = fmaf(yr, t.x, fmaf(yi, t.y, t.z));
    }
#pragma unroll
    for (int b = 0; b < BPS; b++) {
        float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
        for (int s = 0; s < P; s++) { if (((s >> b) & 1) == 0) m0 = fmaxf(m0, u[s]); else m1 = fmaxf(m1, u[s]); }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int s = 0; s < P; s++) { if (((s >> b) & 1) == 0) s0 += __builtin_amdgcn_exp2f(u[s] - m0); else s1 += __builtin_amdgcn_exp2f(u[s] - m1); }
        out[b] = FRONT_LN2 * ((m0 - m1) + (__builtin_amdgcn_logf(s0) - __builtin_amdgcn_logf(s1)));
    }
}

template <int BPS, int U>
__device__ __forceinline__ void demap_symbols(const float2 (&y)[U], const float4 *tab, float (&out)[U][BPS])
{
    constexpr int P = 1 << BPS;
    float u[U][P], mx[U];
#pragma unroll
    for (int i = 0; i < U; i++) mx[i] = -INFINITY;
#if FRONT_ANALYTIC_REF
    // (round 4) the reference exponent without a running maximum over the points: u_s = k (|y|^2 - |y - s|^2) <= k |y|^2, so k |y|^2 bounds every term from above
    // (v_max_f32 per point and symbol is a 4.25-cycle instruction: 32 of them were ~10 % of the 32APSK demapper's issue cycles).  The largest term is then
    // 2^(120 - k d_min^2) instead of 2^120 (d_min: distance to the nearest point, k d_min^2 a few units at the noise levels the decoder works at), the sums keep their
    // relative precision and the per-subset form below takes over a little earlier (|LLR| beyond ~130 instead of ~150).
    const float kq = tab[0].w;
#pragma unroll
    for (int i = 0; i < U; i++) mx[i] = kq * fmaf(y[i].x, y[i].x, y[i].y * y[i].y);
#endif
#pragma unroll
    for (int s = 0; s < P; s++) {
        if ((s & 7) == 0) asm volatile("" ::: "memory");          // at most 8 table rows in flight (registers)
        const float4 t = tab[s];
#pragma unroll
        for (int i = 0; i < U; i++) {
            u[i][s] = fmaf(y[i].x, t.x, fmaf(y[i].y, t.y, t.z));
            if (!FRONT_ANALYTIC_REF) mx[i] = fmaxf(mx[i], u[i][s]);
        }
    }
#pragma unroll
    for (int i = 0; i < U; i++) {
        // relative to 2^-120 of the largest term: the sums stay below 2^(120 + BPS) < 2^128 and a subset keeps full relative
        // precision down to 2^-220 of the overall maximum (|LLR| up to ~150) before the per-subset form is needed
        const float ref = mx[i] - 120.0f;
#pragma unroll
        for (int s = 0; s < P; s++) u[i][s] = __builtin_amdgcn_exp2f(u[i][s] - ref);
        float lo = 1.0f;
#pragma unroll
        for (int b = 0; b < BPS; b++) {
            const int n = P >> b;                      // partial sums left at this level: index bit 0 is bit b of the point
            float s0 = u[i][0], s1 = u[i][1];
#pragma unroll
            for (int j = 1; j < n / 2; j++) { s0 += u[i][2 * j]; s1 += u[i][2 * j + 1]; }
            out[i][b] = FRONT_LN2 * (__builtin_amdgcn_logf(s0) - __builtin_amdgcn_logf(s1));
            lo = fminf(lo, fminf(s0, s1));
#pragma unroll
            for (int j = 0; j < n / 2; j++) u[i][j] = u[i][2 * j] + u[i][2 * j + 1];
        }
        // wave-uniform test first: a real branch around the rare path (a per-lane condition alone is flattened into predicated code)
        if (__builtin_amdgcn_ballot_w64(lo < 0x1p-100f) != 0ull) { if (lo < 0x1p-100f) demap_symbol_far<BPS>(y[i].x, y[i].y, tab, out[i]); }
    }
}

// (round 4) The frame's table on the SCALAR side.  demap_symbols reads the constellation table with one 16-byte LDS broadcast per point and symbol: 32 x 1 KB per
// wave and 32APSK symbol is 0.67 of the LDS pipe's rate at the kernel's speed -- as busy as the vector issue port, so trimming either alone moved nothing (analytic
// reference: -1 %; packed FMAs in round 2: 0).  The table is wave-uniform and constant over the frame: A_s, B_s live in SGPRs (v_readlane once per frame), D_s replicated in
// vector registers (an SGPR operand would make the add a 4.25-cycle instruction), and the inner loop has no LDS access at all:
//     2^(u_s - ref) = exp2( yr A_s + (yi B_s + (120 - k |y|^2)) + D_s ),   pairs of points in the halves of v_pk_fma_f32 / v_pk_add_f32.
typedef float front_v2 __attribute__((ext_vector_type(2)));
template <int BPS>
struct DemapTabS { float A[1 << BPS], B[1 << BPS], D[1 << BPS], k; };
template <int BPS>
__device__ __forceinline__ void demap_tab_scalar(DemapTabS<BPS> &T, const float4 *tab)
{
    constexpr int P = 1 << BPS;
    const float4 t = tab[threadIdx.x & (P - 1)];
#pragma unroll
    for (int s = 0; s < P; s++) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.x), s)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.y), s));
        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.z), s));
        // replicated in vector registers, opaquely: left to itself the compiler keeps the lane-0 copy and re-reads all 2 P values with v_readlane in every iteration of
        // the symbol loop (102 + 48 moves per 32APSK symbol), the operands of the packed instructions being register pairs
        if (FRONT_TAB_VGPR) { asm volatile("v_mov_b32 %0, %1" : "=v"(T.A[s]) : "s"(a)); asm volatile("v_mov_b32 %0, %1" : "=v"(T.B[s]) : "s"(b)); }
        else { T.A[s] = a; T.B[s] = b; }
        asm volatile("v_mov_b32 %0, %1" : "=v"(T.D[s]) : "s"(d));
    }
    T.k = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t.w), 0));
}
template <int BPS, int U>
__device__ __forceinline__ void demap_symbols_s(const float2 (&y)[U], const DemapTabS<BPS> &T, const float4 *tab, float (&out)[U][BPS])
{
    constexpr int P = 1 << BPS;
    float u[U][P];
#pragma unroll
    for (int i = 0; i < U; i++) {
        const float c0 = fmaf(-T.k, fmaf(y[i].x, y[i].x, y[i].y * y[i].y), 120.0f);
#if FRONT_PK
        const front_v2 yr = {y[i].x, y[i].x}, yi = {y[i].y, y[i].y}, cc = {c0, c0};
#pragma unroll
        for (int s = 0; s < P; s += 2) {
            const front_v2 a = {T.A[s], T.A[s + 1]}, b = {T.B[s], T.B[s + 1]}, d = {T.D[s], T.D[s + 1]};
            const front_v2 e = __builtin_elementwise_fma(yr, a, __builtin_elementwise_fma(yi, b, cc)) + d;
            u[i][s] = __builtin_amdgcn_exp2f(e.x); u[i][s + 1] = __builtin_amdgcn_exp2f(e.y);
        }
#else
#pragma unroll
        for (int s = 0; s < P; s++) u[i][s] = __builtin_amdgcn_exp2f(fmaf(y[i].x, T.A[s], fmaf(y[i].y, T.B[s], c0)) + T.D[s]);
#endif
        float lo = 1.0f;
#pragma unroll
        for (int b = 0; b < BPS; b++) {
            const int n = P >> b;                      // partial sums left at this level: index bit 0 is bit b of the point
            float s0 = u[i][0], s1 = u[i][1];
#pragma unroll
            for (int j = 1; j < n / 2; j++) { s0 += u[i][2 * j]; s1 += u[i][2 * j + 1]; }
            out[i][b] = FRONT_LN2 * (__builtin_amdgcn_logf(s0) - __builtin_amdgcn_logf(s1));
            lo = fminf(lo, fminf(s0, s1));
#pragma unroll
            for (int j = 0; j < n / 2; j++) u[i][j] = u[i][2 * j] + u[i][2 * j + 1];
        }
        if (__builtin_amdgcn_ballot_w64(lo < 0x1p-100f) != 0ull) { if (lo < 0x1p-100f) demap_symbol_far<BPS>(y[i].x, y[i].y, tab, out[i]); }
    }
}

__device__ __forceinline__ float block_sum(float v, float *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += red[i];
    return s;
}

// Estimator_DVBS2.hxx:44-56 from the two moment sums.  The divisions, square roots, log10 and 10^x are single hardware instructions
// (v_rcp / v_sqrt / v_log / v_exp, 1 ulp each): the library forms (IEEE division, sqrtf, log10f, powf) expand into sequences that need
// up to ~40 registers of their own at a point where the register-resident kernels hold a whole frame -- they spilled frame symbols for it.
// The estimates end up within ~1e-6 relative of the reference's expressions (compared at 1e-4).
__device__ __forceinline__ void m2m4_finish(float m2, float m4, int n_sym, float code_rate, int bps,
                                            float &sigma, float &ebn0, float &esn0)
{
    const float rn = __builtin_amdgcn_rcpf((float)n_sym);
    m2 *= rn; m4 *= rn;
    const float Se = __builtin_amdgcn_sqrtf(fabsf(2 * m2 * m2 - m4));
    const float Ne = fabsf(m2 - Se);
    esn0 = 10.f * 0.301029995663981f * __builtin_amdgcn_logf(Se * __builtin_amdgcn_rcpf(Ne));      // 10 log10(Se / Ne); Ne = 0: +inf
    if (fabsf(esn0) == INFINITY) esn0 = 100.f;
    sigma = __builtin_amdgcn_sqrtf(__builtin_amdgcn_rcpf(2.0f * __builtin_amdgcn_exp2f(esn0 * 0.332192809488736f)));      // sqrt(1 / (2 10^(esn0 / 10)))
    ebn0 = esn0 - 10.f * 0.301029995663981f * __builtin_amdgcn_logf(code_rate * (float)bps);
}

// separable 2-bit constellation (every reference QPSK mapping): bit b is carried by one axis alone with two levels
// (A0 for bit 0, A1 for bit 1), the other axis cancels out of the ratio and the exact LLR is linear,
//     L_b = c (2 y_ax (A0 - A1) + A1^2 - A0^2),        c = 1 / (2 sigma^2)
// (the pairwise max* form evaluates the same number through four exponentials).  Detected on the host at plan time.
__device__ __forceinline__ void demap_sep2(float2 y, float c, const FrontKParams &p, float *out)
{
    out[0] = c * fmaf(p.sep_ax[0] ? y.y : y.x, p.sep_g[0], p.sep_h[0]);
    out[1] = c * fmaf(p.sep_ax[1] ? y.y : y.x, p.sep_g[1], p.sep_h[1]);
}

// FROM_PL = true : in = PL frames, a7 folded in, sigma estimated unless sigma_in given
// FROM_PL = false: in = XFEC frames, sigma_in required
template <int BPS, bool FROM_PL, bool DEITL>
__global__ void __launch_bounds__(FRONT_THREADS)
front_kernel(const FrontKParams p)
{
    constexpr int U = BPS <= 3 ? 4 : FRONT_U_APSK;          // symbols a lane demaps per read of the constellation table
    __shared__ float4 tab[1 << BPS];
    __shared__ float red[FRONT_THREADS / 64];
    __shared__ float s_sigma;
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n_sym = p.n_sym, n_pil = n_sym / (PL_SLOTS * PL_M);
    const size_t in_stride = FROM_PL ? 2 * (size_t)p.pl_frame : 2 * (size_t)n_sym;
    const float2 *in = reinterpret_cast<const float2 *>((FROM_PL && p.src) ? p.src[f] : p.in + (size_t)f * in_stride);
    float sigma;
    if (FROM_PL && p.sigma_in == nullptr) {
        float m2 = 0.f, m4 = 0.f;
        for (int k = tid; k < n_sym; k += FRONT_THREADS) {
            const int pi = pl_index(k, n_pil);
            const float2 y = in[pi];              // |y| is invariant under the PL derotation
            const float e = y.x * y.x + y.y * y.y;
            m2 += e; m4 += e * e;
        }
        m2 = block_sum(m2, red);
        m4 = block_sum(m4, red);
        float ebn0, esn0;
        m2m4_finish(m2, m4, n_sym, p.code_rate, p.bps, sigma, ebn0, esn0);
        if (tid == 0 && p.est && blockIdx.y == 0) { p.est[3 * f] = sigma; p.est[3 * f + 1] = ebn0; p.est[3 * f + 2] = esn0; }
    } else {
        if (tid == 0) s_sigma = p.sigma_in[f];
        __syncthreads();
        sigma = s_sigma;
        if (tid == 0 && p.est && blockIdx.y == 0) { p.est[3 * f] = sigma; p.est[3 * f + 1] = 0.f; p.est[3 * f + 2] = 0.f; }
    }
    const float inv2s2 = 1.0f / (2.0f * sigma * sigma);
    if (tid < (1 << BPS)) tab[tid] = demap_table_entry(p.cstl, tid, inv2s2);
    __syncthreads();
    constexpr bool TS = FRONT_TAB_SCALAR && BPS >= 3;
    DemapTabS<TS ? BPS : 0> T;
    if constexpr (TS) demap_tab_scalar<BPS>(T, tab);
    float *llr = p.llr + (size_t)f * n_sym * BPS;
    const int n_rows = (n_sym * BPS) / (p.itl_cols > 1 ? p.itl_cols : 1);
    const bool sep = BPS == 2 && p.sep != 0;
    const bool st_fast = !DEITL || p.itl_cols <= 1 || p.itl_cols == BPS;
    const bool st_rows = DEITL && p.itl_cols == BPS;
    const uint32_t kstride = st_rows ? 4u : 4u * BPS;
    uint32_t boff[BPS];
#pragma unroll
    for (int b = 0; b < BPS; b++) boff[b] = st_rows ? (uint32_t)((p.itl_order == DVBS2HIP_ITL_TOP_LEFT ? b : BPS - 1 - b) * n_rows) * 4u : 4u * b;
    const __amdgpu_buffer_rsrc_t rllr = __builtin_amdgcn_make_buffer_rsrc(llr, 0, n_sym * BPS * 4, 0x00020000);
    // the symbols of iteration i + 1 are requested before those of iteration i are demapped (the arithmetic of one
    // iteration is about as long as an HBM round trip, and a lane has nothing else to overlap it with)
    float2 yn[U];
    int Rn[U];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int k = k0 + i * FRONT_THREADS;
            yn[i] = make_float2(0.f, 0.f); Rn[i] = 0;
            if (k < n_sym) {
                if (FROM_PL) { const int pi = pl_index(k, n_pil); yn[i] = in[pi]; Rn[i] = p.pl_seq[pi - PL_M]; }
                else yn[i] = in[k];
            }
        }
    };
    // (round 5) small batches: a frame's symbols are dealt to gridDim.y workgroups in pieces of slice_chunk (a multiple of what a workgroup takes per trip), every one of which forms
    // sigma from the whole frame for itself -- one workgroup walking a short 32APSK frame alone was 42 us of the 250 us a one-frame call sequence takes (tools/r05_latency_trace.sh)
    const int k_lo = p.slice_chunk ? (int)blockIdx.y * p.slice_chunk : 0, k_hi = p.slice_chunk ? min(n_sym, k_lo + p.slice_chunk) : n_sym;
    fetch(k_lo + tid);
    for (int k0 = k_lo + tid; k0 < k_hi; k0 += U * FRONT_THREADS) {
        float2 y[U];
#pragma unroll
        for (int i = 0; i < U; i++) y[i] = FROM_PL ? pl_derotate(yn[i], Rn[i]) : yn[i];
        fetch(k0 + U * FRONT_THREADS);
        float out[U][BPS];
        if (sep) {
#pragma unroll
            for (int i = 0; i < U; i++) demap_sep2(y[i], inv2s2, p, out[i]);
        } else if constexpr (TS) demap_symbols_s<BPS, U>(y, T, tab, out);
        else demap_symbols<BPS, U>(y, tab, out);
#pragma unroll
        for (int i = 0; i < U; i++) {
            const int k = k0 + i * FRONT_THREADS;
            if (k >= n_sym) continue;
            if (FRONT_BUF_ST && st_fast) {
                // LLR b of symbol k sits at  k * kstride + boff[b]  (no interleaver: k bps + b; as many columns as bits: column (b or cols - 1 - b), row k): one vector offset
                // per symbol, the bit's part in the instruction's scalar offset -- the generic index below costs three wave-uniform branches and a 64-bit address per LLR
#pragma unroll
                for (int b = 0; b < BPS; b++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[i][b]), rllr, (uint32_t)k * kstride, boff[b], 0);
                continue;
            }
#pragma unroll
            for (int b = 0; b < BPS; b++) {
                const int dst = DEITL ? deitl_index(k, b, BPS, p.itl_cols, p.itl_order, n_rows) : k * BPS + b;
                llr[dst] = out[i][b];
            }
        }
    }
}

// ---- the fused front end for QPSK / 8PSK with the frame held in REGISTERS between its two sweeps: 1024 lanes per frame,
// SPT symbols per lane, so the PL frame crosses the fabric once (the moments of a6 need the whole frame before the
// first LLR of a3 can be formed; the two-sweep kernel above re-reads it from L2 / Infinity Cache), the loads of a lane
// are all in flight together, and QPSK LLR pairs leave as one 8-byte store per lane.
constexpr int FRONT_WIDE = 1024;
typedef float front_f2 __attribute__((ext_vector_type(2)));      // a type the non-temporal builtins accept
template <int BPS, int SPT, bool SEP, int WIDE = FRONT_WIDE>      // SEP: separable 2-bit constellation, linear LLRs (no general demapper compiled in); WIDE lanes per frame
__global__ void __launch_bounds__(WIDE, 4)      // four waves per SIMD: one 1024-lane or two 512-lane workgroups per CU, 128 registers
front_reg_kernel(const FrontKParams p)
{
    __shared__ float4 tab[1 << BPS];
    __shared__ float red[2][WIDE / 64];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n_sym = p.n_sym, n_pil = n_sym / (PL_SLOTS * PL_M);
    const float2 *in = reinterpret_cast<const float2 *>(p.src ? p.src[f] : p.in + (size_t)f * 2 * (size_t)p.pl_frame);
    float2 y[SPT];
    // 32-bit offsets through buffer descriptors (a 64-bit address per load in flight would not fit the register budget);
    // an offset past the frame returns zero, which is what the padding lanes have to hold
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in), 0, 8 * p.pl_frame, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.pl_seq), 0, p.pl_frame, 0x00020000);
#pragma clang loop unroll(full)
    for (int i = 0; i < SPT; i++) {
        const int k = tid + i * WIDE;
        const int pi = k < n_sym ? pl_index(k, n_pil) : p.pl_frame;
        const front_f2 v = __builtin_bit_cast(front_f2, __builtin_amdgcn_raw_buffer_load_b64(rin, 8 * pi, 0, 2));      // nt: read once
        y[i] = make_float2(v.x, v.y);
        if (SPT > 8 && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // offsets formed four at a time, not all SPT of them ahead of the first load (registers)
    }
    float sigma;
    if (p.sigma_in == nullptr) {
        float m2 = 0.f, m4 = 0.f;                  // |y| is invariant under the PL derotation; padding lanes hold zeros
#pragma clang loop unroll(full)
        for (int i = 0; i < SPT; i++) {
            const float e = y[i].x * y[i].x + y[i].y * y[i].y; m2 += e; m4 += e * e;
            if (SPT > 8 && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four squares at a time, not SPT of them beside the frame (registers)
        }
        for (int o = 32; o > 0; o >>= 1) { m2 += __shfl_xor(m2, o); m4 += __shfl_xor(m4, o); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = m2; red[1][tid >> 6] = m4; }
        __syncthreads();
        m2 = 0.f; m4 = 0.f;
        for (int i = 0; i < WIDE / 64; i++) { m2 += red[0][i]; m4 += red[1][i]; }
        float ebn0, esn0;
        m2m4_finish(m2, m4, n_sym, p.code_rate, p.bps, sigma, ebn0, esn0);
        if (tid == 0 && p.est) { p.est[3 * f] = sigma; p.est[3 * f + 1] = ebn0; p.est[3 * f + 2] = esn0; }
    } else {
        sigma = p.sigma_in[f];
        if (tid == 0 && p.est) { p.est[3 * f] = sigma; p.est[3 * f + 1] = 0.f; p.est[3 * f + 2] = 0.f; }
    }
    const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sigma * sigma);
    if (!SEP) {
        if (tid < (1 << BPS)) tab[tid] = demap_table_entry(p.cstl, tid, inv2s2);
        __syncthreads();
    }
    constexpr bool TS = FRONT_TAB_SCALAR && FRONT_REG_TS && BPS >= 3 && !SEP;
    DemapTabS<TS ? BPS : 0> T;
    if constexpr (TS) demap_tab_scalar<BPS>(T, tab);
    float *llr = p.llr + (size_t)f * n_sym * BPS;
    const int n_rows = (n_sym * BPS) / (p.itl_cols > 1 ? p.itl_cols : 1);
    const bool pairs = BPS == 2 && p.itl_cols <= 1;
    const bool st_fast = FRONT_REG_TS && !pairs && (p.itl_cols <= 1 || p.itl_cols == BPS), st_rows = p.itl_cols == BPS;
    const uint32_t kstride = st_rows ? 4u : 4u * BPS;
    uint32_t boff[BPS];
#pragma unroll
    for (int b = 0; b < BPS; b++) boff[b] = st_rows ? (uint32_t)((p.itl_order == DVBS2HIP_ITL_TOP_LEFT ? b : BPS - 1 - b) * n_rows) * 4u : 4u * b;
    const __amdgpu_buffer_rsrc_t rllr = __builtin_amdgcn_make_buffer_rsrc(llr, 0, n_sym * BPS * 4, 0x00020000);
    constexpr bool sep = SEP;
    constexpr int U = 1;                           // the frame itself fills the registers
    static_assert(SPT % U == 0, "symbols per lane come in groups of U");
#pragma clang loop unroll(full)
    for (int i0 = 0; i0 < SPT; i0 += U) {
        asm volatile("" ::: "memory");             // one group at a time: nothing of the next one is hoisted into registers
        if (tid + i0 * WIDE >= n_sym) continue;
        float2 yy[U];
        float out[U][BPS];
#pragma clang loop unroll(full)
        for (int i = 0; i < U; i++) {
            const int k = tid + (i0 + i) * WIDE;
            const int R = (int)__builtin_amdgcn_raw_buffer_load_b8(rsq, (k < n_sym ? pl_index(k, n_pil) : PL_M) - PL_M, 0, 0);      // L2-resident table
            yy[i] = pl_derotate(y[i0 + i], R);
        }
        if constexpr (sep) {
#pragma clang loop unroll(full)
            for (int i = 0; i < U; i++) demap_sep2(yy[i], inv2s2, p, out[i]);
        } else if constexpr (TS) demap_symbols_s<BPS, U>(yy, T, tab, out);
        else demap_symbols<BPS, U>(yy, tab, out);
#pragma clang loop unroll(full)
        for (int i = 0; i < U; i++) {
            const int k = tid + (i0 + i) * WIDE;
            if (k >= n_sym) continue;
            if (st_fast) {
#pragma clang loop unroll(full)
                for (int b = 0; b < BPS; b++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[i][b]), rllr, (uint32_t)k * kstride, boff[b], 0);
            } else if (pairs) { front_f2 v; v.x = out[i][0]; v.y = out[i][BPS - 1]; __builtin_nontemporal_store(v, reinterpret_cast<front_f2 *>(llr) + k); }
            else {
#pragma clang loop unroll(full)
                for (int b = 0; b < BPS; b++) llr[deitl_index(k, b, BPS, p.itl_cols, p.itl_order, n_rows)] = out[i][b];
            }
        }
    }
}

// ---- the same for the separable QPSK mappings without an interleaver (every reference QPSK MODCOD), two neighbouring symbols per lane and
// access: a pair never straddles a pilot block (16 slots = 1440 symbols, even), so it is 16 bytes of the PL frame, two bytes of the
// PL sequence and 16 bytes of LLRs -- half the vector-memory instructions of front_reg_kernel for the same bytes (same-box: QPSK-N
// 4096 frames 0.451 -> see DESIGN section 4).  PPT pairs per lane; needs 16-byte aligned sockets and an even number of symbols.
typedef float front_f4 __attribute__((ext_vector_type(4)));
template <int PPT>
__global__ void __launch_bounds__(FRONT_WIDE)
front_reg2_kernel(const FrontKParams p)
{
    __shared__ float red[2][FRONT_WIDE / 64];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n_sym = p.n_sym, n_pil = n_sym / (PL_SLOTS * PL_M), n_pairs = n_sym / 2;
    const float2 *in = reinterpret_cast<const float2 *>(p.src ? p.src[f] : p.in + (size_t)f * 2 * (size_t)p.pl_frame);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(in), 0, 8 * p.pl_frame, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.pl_seq), 0, p.pl_frame, 0x00020000);
    front_f4 y[PPT];
    int pio[PPT];
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int j = tid + i * FRONT_WIDE;                       // pair j = symbols 2 j, 2 j + 1
        pio[i] = j < n_pairs ? pl_index(2 * j, n_pil) : p.pl_frame;      // past the frame: the load returns zeros (padding lanes)
        y[i] = __builtin_bit_cast(front_f4, __builtin_amdgcn_raw_buffer_load_b128(rin, 8 * pio[i], 0, 2));      // nt: read once
    }
    float sigma;
    if (p.sigma_in == nullptr) {
        // the moments in the order of front_reg_kernel's lanes would need the same symbol-to-lane map; the sums differ from it in the last
        // bits (summation order), as any two estimators' do (tests/test_front_gpu.py: 1e-4 relative against the oracle's serial sum)
        float m2 = 0.f, m4 = 0.f;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const float e0 = y[i].x * y[i].x + y[i].y * y[i].y, e1 = y[i].z * y[i].z + y[i].w * y[i].w;
            m2 += e0; m4 += e0 * e0; m2 += e1; m4 += e1 * e1;
        }
        for (int o = 32; o > 0; o >>= 1) { m2 += __shfl_xor(m2, o); m4 += __shfl_xor(m4, o); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = m2; red[1][tid >> 6] = m4; }
        __syncthreads();
        m2 = 0.f; m4 = 0.f;
        for (int i = 0; i < FRONT_WIDE / 64; i++) { m2 += red[0][i]; m4 += red[1][i]; }
        float ebn0, esn0;
        m2m4_finish(m2, m4, n_sym, p.code_rate, p.bps, sigma, ebn0, esn0);
        if (tid == 0 && p.est) { p.est[3 * f] = sigma; p.est[3 * f + 1] = ebn0; p.est[3 * f + 2] = esn0; }
    } else {
        sigma = p.sigma_in[f];
        if (tid == 0 && p.est) { p.est[3 * f] = sigma; p.est[3 * f + 1] = 0.f; p.est[3 * f + 2] = 0.f; }
    }
    const float inv2s2 = __builtin_amdgcn_rcpf(2.0f * sigma * sigma);
    front_f4 *llr = reinterpret_cast<front_f4 *>(p.llr + (size_t)f * n_sym * 2);
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int j = tid + i * FRONT_WIDE;
        if (j >= n_pairs) continue;
        const uint32_t R = __builtin_amdgcn_raw_buffer_load_b16(rsq, pio[i] - PL_M, 0, 0);      // two bytes of the L2-resident PL sequence
        const float2 a = pl_derotate(make_float2(y[i].x, y[i].y), (int)(R & 0xffu)), b = pl_derotate(make_float2(y[i].z, y[i].w), (int)(R >> 8));
        float oa[2], ob[2];
        demap_sep2(a, inv2s2, p, oa);
        demap_sep2(b, inv2s2, p, ob);
        __builtin_nontemporal_store(front_f4{oa[0], oa[1], ob[0], ob[1]}, llr + j);
    }
}

static bool front_reg_try(const FrontKParams &p, hipStream_t s)
{
    if (getenv("DVBS2HIP_FRONT_TWO_SWEEP")) return false;
    dim3 g(p.n_frames), b(FRONT_WIDE);
    const bool small = p.n_sym <= 8 * FRONT_WIDE, mid = p.n_sym <= 22 * FRONT_WIDE, big = p.n_sym <= 32 * FRONT_WIDE;
    const bool pair_ok = p.bps == 2 && p.sep && p.itl_cols <= 1 && p.n_sym % 2 == 0 && p.pl_frame % 2 == 0 && !getenv("DVBS2HIP_FRONT_SINGLE") &&
                         ((reinterpret_cast<uintptr_t>(p.src ? nullptr : p.in) | reinterpret_cast<uintptr_t>(p.llr)) & 15) == 0;      // (a located frame starts on 8 bytes: the pair kernel reads it through a buffer descriptor, which takes any dword address)
    if (pair_ok && small) hipLaunchKernelGGL((front_reg2_kernel<4>), g, b, 0, s, p);
    else if (pair_ok && big) hipLaunchKernelGGL((front_reg2_kernel<16>), g, b, 0, s, p);
    // one symbol per lane and access: short frames (8 symbols per lane) and the 8PSK normal frame (21600 symbols: 22 per lane = 44 registers
    // of frame, which leaves the demapper its own; round 2's 32-per-lane 8PSK form spilled frame symbols and is gone).  A separable QPSK normal frame that
    // cannot take the pair kernel (unaligned sockets) still runs the 32-per-lane QPSK form below (no spills since the estimator's finish lost its powf / log10f);
    // a non-separable mapping goes to the two-sweep kernel
    else if (p.bps == 2 && p.sep && small) hipLaunchKernelGGL((front_reg_kernel<2, 8, true>), g, b, 0, s, p);
    else if (p.bps == 2 && p.sep && big) hipLaunchKernelGGL((front_reg_kernel<2, 32, true>), g, b, 0, s, p);
    else if (p.bps == 2 && small) hipLaunchKernelGGL((front_reg_kernel<2, 8, false>), g, b, 0, s, p);
    else if (p.bps == 3 && small) hipLaunchKernelGGL((front_reg_kernel<3, 8, false>), g, b, 0, s, p);
    else if (p.bps == 3 && mid) hipLaunchKernelGGL((front_reg_kernel<3, 22, false>), g, b, 0, s, p);
    // (round 4: the APSK frames in registers as well -- front_reg_kernel<4, 4>, <4, 16>, <5, 4> -- measured 0.398 against 0.398 ms (16APSK-N), 0.098 against 0.093 (16APSK-S),
    // 0.136 against 0.121 (32APSK-S): with sigma given they read the frame once either way, and the two-sweep kernel overlaps eight workgroups per CU)
    else return false;
    return true;
}

template <bool FROM_PL, bool DEITL>
static hipError_t front_dispatch(const FrontKParams &p_in, hipStream_t s)
{
    FrontKParams p = p_in;
    p.slice_chunk = 0;
    dim3 g(p.n_frames), b(FRONT_THREADS);
    if (p.n_frames < 512) {      // fewer workgroups than the chip holds: several per frame
        const int trip = (p.bps <= 3 ? 4 : FRONT_U_APSK) * FRONT_THREADS;      // symbols a workgroup takes per trip of its loop (front_kernel's U)
        int slices = (p.n_sym + trip - 1) / trip;
        const int cap = 1024 / p.n_frames;
        if (slices > cap) slices = cap;
        if (slices > 1) { p.slice_chunk = ((p.n_sym + slices - 1) / slices + trip - 1) / trip * trip; g.y = (unsigned)((p.n_sym + p.slice_chunk - 1) / p.slice_chunk); }
    }
    switch (p.bps) {
        case 1: hipLaunchKernelGGL((front_kernel<1, FROM_PL, DEITL>), g, b, 0, s, p); break;
        case 2: hipLaunchKernelGGL((front_kernel<2, FROM_PL, DEITL>), g, b, 0, s, p); break;
        case 3: hipLaunchKernelGGL((front_kernel<3, FROM_PL, DEITL>), g, b, 0, s, p); break;
        case 4: hipLaunchKernelGGL((front_kernel<4, FROM_PL, DEITL>), g, b, 0, s, p); break;
        case 5: hipLaunchKernelGGL((front_kernel<5, FROM_PL, DEITL>), g, b, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t front_rx_launch(FrontKParams p, hipStream_t s)
{
    if (front_reg_try(p, s)) return hipGetLastError();
    return front_dispatch<true, true>(p, s);
}
hipError_t demod_launch(FrontKParams p, bool deinterleave, hipStream_t s)
{
    return deinterleave ? front_dispatch<false, true>(p, s) : front_dispatch<false, false>(p, s);
}

// ---------------------------------------------------------------- a4 stand-alone
__global__ void deinterleave_kernel(const float *itl, float *nat, int N, int cols, int order, int n_rows)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int dst = i;
    if (cols > 1) {
        // the DVB-S2 column counts divide by a constant (multiply + shift); any other count takes the general division
        const int row = cols == 3 ? i / 3 : cols == 4 ? i / 4 : cols == 5 ? i / 5 : cols == 2 ? i / 2 : i / cols, j = i - row * cols;
        dst = (order == DVBS2HIP_ITL_TOP_LEFT ? j : cols - 1 - j) * n_rows + row;
    }
    nat[(size_t)f * N + dst] = itl[(size_t)f * N + i];
}
hipError_t deinterleave_launch(const float *itl, float *nat, int N, int cols, int order, int F, hipStream_t s)
{
    const int n_rows = N / (cols > 1 ? cols : 1);
    hipLaunchKernelGGL(deinterleave_kernel, dim3((N + 255) / 256, F), dim3(256), 0, s, itl, nat, N, cols, order, n_rows);
    return hipGetLastError();
}

// ---------------------------------------------------------------- a6 stand-alone
__global__ void __launch_bounds__(FRONT_THREADS)
estimate_kernel(const float *x, float *sig, float *ebn0o, float *esn0o, int n_sym, float code_rate, int bps)
{
    __shared__ float red[FRONT_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float2 *in = reinterpret_cast<const float2 *>(x + (size_t)f * 2 * n_sym);
    float m2 = 0.f, m4 = 0.f;
    for (int k = tid; k < n_sym; k += FRONT_THREADS) {
        const float2 y = in[k];
        const float e = y.x * y.x + y.y * y.y;
        m2 += e; m4 += e * e;
    }
    m2 = block_sum(m2, red);
    m4 = block_sum(m4, red);
    float sigma, ebn0, esn0;
    m2m4_finish(m2, m4, n_sym, code_rate, bps, sigma, ebn0, esn0);
    if (tid == 0) { sig[f] = sigma; ebn0o[f] = ebn0; esn0o[f] = esn0; }
}
hipError_t estimate_launch(const float *x, float *sig, float *ebn0, float *esn0, int n_sym, float code_rate,
                           int bps, int F, hipStream_t s)
{
    hipLaunchKernelGGL(estimate_kernel, dim3(F), dim3(FRONT_THREADS), 0, s, x, sig, ebn0, esn0, n_sym, code_rate, bps);
    return hipGetLastError();
}

// ---------------------------------------------------------------- a7 stand-alone
__global__ void pl_descramble_kernel(const float *in, float *out, const uint8_t *seq, int pl_frame)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl_frame) return;
    const float2 *src = reinterpret_cast<const float2 *>(in + (size_t)f * 2 * pl_frame);
    float2 *dst = reinterpret_cast<float2 *>(out + (size_t)f * 2 * pl_frame);
    float2 y = src[i];
    if (i >= PL_M) y = pl_derotate(y, seq[i - PL_M]);
    dst[i] = y;
}
hipError_t pl_descramble_launch(const float *in, float *out, const uint8_t *seq, int pl_frame, int F, hipStream_t s)
{
    hipLaunchKernelGGL(pl_descramble_kernel, dim3((pl_frame + 255) / 256, F), dim3(256), 0, s, in, out, seq, pl_frame);
    return hipGetLastError();
}

__global__ void remove_plh_kernel(const float *in, float *out, int n_sym, int pl_frame)
{
    const int f = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sym) return;
    const float2 *src = reinterpret_cast<const float2 *>(in + (size_t)f * 2 * pl_frame);
    float2 *dst = reinterpret_cast<float2 *>(out + (size_t)f * 2 * n_sym);
    dst[k] = src[pl_index(k, n_sym / (PL_SLOTS * PL_M))];
}
hipError_t remove_plh_launch(const float *in, float *out, int n_sym, int pl_frame, int F, hipStream_t s)
{
    hipLaunchKernelGGL(remove_plh_kernel, dim3((n_sym + 255) / 256, F), dim3(256), 0, s, in, out, n_sym, pl_frame);
    return hipGetLastError();
}

// ---------------------------------------------------------------- a8 stand-alone
__global__ void bb_descramble_kernel(const int32_t *in, int32_t *out, const uint32_t *prbs, int K)
{
    const int f = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int32_t fb = (int32_t)((prbs[k >> 5] >> (k & 31)) & 1u);
    out[(size_t)f * K + k] = (in[(size_t)f * K + k] + fb) % 2;        // Scrambler_BB.hxx:63
}
hipError_t bb_descramble_launch(const int32_t *in, int32_t *out, const uint32_t *prbs, int K, int F, hipStream_t s)
{
    hipLaunchKernelGGL(bb_descramble_kernel, dim3((K + 255) / 256, F), dim3(256), 0, s, in, out, prbs, K);
    return hipGetLastError();
}

// bit errors of one frame seen by lane tid of its workgroup: four bits (16 bytes of each socket, read once: non-temporal) per access where the
// frame allows it
typedef int mon_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int monitor_frame_errors(const int32_t *__restrict__ u, const int32_t *__restrict__ v, int K, int tid)
{
    int be = 0;
    if ((K & 3) == 0 && ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        const mon_i4 *u4 = reinterpret_cast<const mon_i4 *>(u), *v4 = reinterpret_cast<const mon_i4 *>(v);
        for (int k = tid; k < K / 4; k += FRONT_THREADS) {
            const mon_i4 a = __builtin_nontemporal_load(u4 + k), b = __builtin_nontemporal_load(v4 + k);
            be += (a.x != b.x) + (a.y != b.y) + (a.z != b.z) + (a.w != b.w);
        }
    } else
        for (int k = tid; k < K; k += FRONT_THREADS) be += (u[k] != v[k]) ? 1 : 0;
    return be;
}

// ---------------------------------------------------------------- a9
__global__ void __launch_bounds__(FRONT_THREADS)
monitor_kernel(const int32_t *U, const int32_t *V, unsigned long long *ctr, int K)
{
    __shared__ int red[FRONT_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int be0 = monitor_frame_errors(U + (size_t)f * K, V + (size_t)f * K, K, tid);
    int be = be0;
    for (int o = 32; o > 0; o >>= 1) be += __shfl_xor(be, o);
    if ((tid & 63) == 0) red[tid >> 6] = be;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int i = 0; i < FRONT_THREADS / 64; i++) tot += red[i];
        atomicAdd(&ctr[0], 1ull);
        if (tot) { atomicAdd(&ctr[1], (unsigned long long)tot); atomicAdd(&ctr[2], 1ull); }
    }
}
hipError_t monitor_launch(const int32_t *U, const int32_t *V, unsigned long long *ctr, int K, int F, hipStream_t s)
{
    hipLaunchKernelGGL(monitor_kernel, dim3(F), dim3(FRONT_THREADS), 0, s, U, V, ctr, K);
    return hipGetLastError();
}

// check_errors2 (aff3ct Monitor_BFER, task bound at RX/main_sched.cpp:222-223,244-247): the counters AFTER every frame of the
// call leave through per-frame sockets.  Pass 1: bit errors of every frame (one workgroup per frame, no atomics); pass 2: one
// workgroup scans them from the counters the call started with, writes the five sockets and the new counters.
__global__ void __launch_bounds__(FRONT_THREADS)
monitor_be_kernel(const int32_t *U, const int32_t *V, int32_t *be_out, int K)
{
    __shared__ int red[FRONT_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int be0 = monitor_frame_errors(U + (size_t)f * K, V + (size_t)f * K, K, tid);
    int be = be0;
    for (int o = 32; o > 0; o >>= 1) be += __shfl_xor(be, o);
    if ((tid & 63) == 0) red[tid >> 6] = be;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int i = 0; i < FRONT_THREADS / 64; i++) tot += red[i];
        be_out[f] = tot;
    }
}

__global__ void __launch_bounds__(1024)
monitor_scan_kernel(const int32_t *be_f, unsigned long long *ctr, long long *FRA, int32_t *BE, int32_t *FE, float *BER, float *FER, int K, int F)
{
    __shared__ unsigned long long s_be[1024];
    __shared__ unsigned int s_fe[1024];
    const int tid = threadIdx.x;
    const int per = (F + 1023) / 1024, f0 = tid * per, f1 = min(F, f0 + per);
    unsigned long long be = 0; unsigned int fe = 0;
    for (int f = f0; f < f1; f++) { be += (unsigned)be_f[f]; fe += be_f[f] != 0; }
    s_be[tid] = be; s_fe[tid] = fe;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                    // inclusive scan over the 1024 segments
        unsigned long long b = tid >= o ? s_be[tid - o] : 0ull; unsigned int e = tid >= o ? s_fe[tid - o] : 0u;
        __syncthreads();
        s_be[tid] += b; s_fe[tid] += e;
        __syncthreads();
    }
    const unsigned long long fra0 = ctr[0], be0 = ctr[1], fe0 = ctr[2];
    unsigned long long cb = be0 + (tid ? s_be[tid - 1] : 0ull), ce = fe0 + (tid ? s_fe[tid - 1] : 0u);
    for (int f = f0; f < f1; f++) {
        cb += (unsigned)be_f[f]; ce += be_f[f] != 0;
        const unsigned long long fra = fra0 + f + 1;
        if (FRA) FRA[f] = (long long)fra;
        if (BE) BE[f] = (int32_t)cb;
        if (FE) FE[f] = (int32_t)ce;
        // Monitor_BFER::get_ber / get_fer (aff3ct): with no error yet they report the bound 1 / n_fra [/ K], not zero
        if (FER) FER[f] = cb ? (float)ce / (float)fra : 1.f / (float)fra;
        if (BER) BER[f] = cb ? (float)cb / (float)fra / (float)K : 1.f / (float)fra / (float)K;
    }
    __syncthreads();
    if (tid == 1023) { ctr[0] = fra0 + F; ctr[1] = be0 + s_be[1023]; ctr[2] = fe0 + s_fe[1023]; }
}
hipError_t monitor2_launch(const int32_t *U, const int32_t *V, unsigned long long *ctr, int32_t *be_tmp, long long *FRA, int32_t *BE, int32_t *FE, float *BER,
                           float *FER, int K, int F, hipStream_t s)
{
    hipLaunchKernelGGL(monitor_be_kernel, dim3(F), dim3(FRONT_THREADS), 0, s, U, V, be_tmp, K);
    hipLaunchKernelGGL(monitor_scan_kernel, dim3(1), dim3(1024), 0, s, be_tmp, ctr, FRA, BE, FE, BER, FER, K, F);
    return hipGetLastError();
}

}  // namespace dvbs2
