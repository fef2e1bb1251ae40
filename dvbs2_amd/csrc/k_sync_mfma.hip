// N4 -- the two correlators of the frame synchronizer on the matrix cores, gfx950.
//
// Synchronizer_frame_DVBS2_fast::_synchronize1 (/root/reference src/common/Module/Synchronizer/Synchronizer_frame/
// Synchronizer_frame_DVBS2_fast.cpp:132-150): d[i] = x[i-1] conj(x[i]), cor_PLSC = FIR(d, 64 real taps, every other one 0),
// cor_SOF = FIR(d, 25 real taps), all taps +-1.  The vector kernel of k_sync.hip spends 57 packed additions per sample on them
// and is bound by that (profiles/r02_kernels_pmc.md); a +-1 tap is exact in bf16, so the same sums are the banded-Toeplitz
// products of k_fir_mfma.hip with ONE tap part instead of three:
//     cor[16 a + i] = sum_{k=0}^{95} A[i][k] d[16 (a - 5) + k],     A[i][k] = brev81[k - i]
// with the differential samples split exactly into three bf16 terms (8 + 8 + 8 significant bits) while they are staged into
// LDS, the three products accumulated smallest first in the MFMA's fp32 accumulator.  Every product is exact, so what differs
// from the reference's chain of fp32 fmas is the order of the fp32 additions only (1e-7 relative; the parity bar of the
// correlation sockets is 1e-5 per tap, tests/test_sync_gpu.py).  The 25 SOF taps occupy k = 56 + i .. 80 + i: K steps 1 and 2.
//
// FUSED (the one-task _synchronize, :46-128): the metric pairs cor_PLSC[o] with cor_SOF[o - 64] (:236), i.e. with the SOF
// correlation of the windows FOUR BLOCKS earlier -- the same lane map, sample fragments 64 entries further back, no exchange
// between lanes; only m[o] = max(|plsc + sof|, |sof - plsc|) leaves the chip (8 B in, 4 B out per sample).  The first 64
// samples of a call take cor_SOF from the handle's history (sofh), the cor_SOF of the call's last 64 samples is formed by the
// tiles that hold them (same fragments, same order of operations as the two-task form: identical bits) and goes to sofh_out.
//
// Lane maps as in k_fir_mfma.hip: lane (c = l & 15, g = l >> 4) holds samples 8 g .. 8 g + 7 of window c (A operand) and taps
// 8 g .. 8 g + 7 of output column c (B operand) per 32-wide K step; accumulator r is output c of block 4 g + r.
#include "dvbs2hip_internal.h"
#include <vector>
#include <cstring>

namespace dvbs2 {

constexpr int SM_THREADS = 256;
constexpr int SM_TILE = 2048;                      // outputs per workgroup = 8 MFMA tiles of 256
constexpr int SM_HALO = 144;                       // 80 (band layout of fir_mfma_afrag) + 64 (SOF_PLSC_delay)
constexpr int SM_NS = SM_TILE + SM_HALO;           // staged differential samples; plane index j <-> d[blk0 - 144 + j]
constexpr int SM_PLANE = SM_NS;
constexpr int SM_H = 64;                           // samples of x kept from the previous call (k_sync.hip SY_H)
static_assert(SM_PLANE % 8 == 0 && SM_HALO % 16 == 0 && SM_TILE % (2 * SM_THREADS) == 0 && SM_HALO / 2 <= SM_THREADS, "whole fragments, whole passes");
constexpr int SM_NPASS = SM_TILE / (2 * SM_THREADS);

typedef __bf16 sm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sm_f32x4 __attribute__((ext_vector_type(4)));
typedef float sm_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sm_pk(float a, float b)      // v_cvt_pk_bf16_f32, round to nearest even
{
    sm_f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sm_bf16x2));
}
__device__ __forceinline__ void sm_split(float a, float b, uint32_t &p1, uint32_t &p2, uint32_t &p3)
{
    p1 = sm_pk(a, b);
    a -= __uint_as_float(p1 << 16); b -= __uint_as_float(p1 & 0xffff0000u);
    p2 = sm_pk(a, b);
    a -= __uint_as_float(p2 << 16); b -= __uint_as_float(p2 & 0xffff0000u);
    p3 = sm_pk(a, b);
}
// the stream before this call = the handle's 64-sample memory, nothing before that and nothing after the end
__device__ __forceinline__ float2 sm_fetch(const float2 *__restrict__ x, const float2 *__restrict__ xh, long long n_total, long long g)
{
    float2 v = make_float2(0.f, 0.f);
    if (g < 0) { if (g >= -SM_H) v = xh[SM_H + g]; }
    else if (g < n_total) v = x[g];
    return v;
}
// d of two consecutive samples (a = x[g - 1], b = x[g], c = x[g + 1]; :138-142, the previous sample times the conjugate of this one),
// split and stored at index o of the six planes [part][re | im]
__device__ __forceinline__ void sm_stage(uint16_t *lds, int o, float2 a, float2 b, float2 c)
{
    const float re0 = a.x * b.x + a.y * b.y, im0 = a.y * b.x - a.x * b.y;
    const float re1 = b.x * c.x + b.y * c.y, im1 = b.y * c.x - b.x * c.y;
    uint32_t p1, p2, p3;
    sm_split(re0, re1, p1, p2, p3);
    *reinterpret_cast<uint32_t *>(lds + 0 * SM_PLANE + o) = p1;
    *reinterpret_cast<uint32_t *>(lds + 2 * SM_PLANE + o) = p2;
    *reinterpret_cast<uint32_t *>(lds + 4 * SM_PLANE + o) = p3;
    sm_split(im0, im1, p1, p2, p3);
    *reinterpret_cast<uint32_t *>(lds + 1 * SM_PLANE + o) = p1;
    *reinterpret_cast<uint32_t *>(lds + 3 * SM_PLANE + o) = p2;
    *reinterpret_cast<uint32_t *>(lds + 5 * SM_PLANE + o) = p3;
}

// One workgroup = 2048 outputs.  EDGE = false is the body of every workgroup whose samples, halo included, lie inside this call's stream and
// that has nothing to do with the memories (all but the first and the last one or two): no range test per lane, uniform base pointers with
// 32-bit lane offsets.  The kernel is bound by the instructions it issues (staging ~45 per pair of samples, 30 matrix products and ~45 vector
// instructions per 256 outputs; profiles/), not by the 12 bytes per sample it moves, so the tests the edges need stay out of the common path.
template <bool FUSED, bool EDGE>
__device__ __forceinline__ void sm_body(uint16_t *lds, const float2 *__restrict__ x, const float2 *__restrict__ xh, const uint4 *__restrict__ frag,
                                        const float2 *__restrict__ sofh, float2 *__restrict__ sofh_out, float *__restrict__ corr, float2 *__restrict__ cor_sof,
                                        float2 *__restrict__ cor_plsc, long long n_total, long long blk0)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float2 *xb = x + blk0;                                     // uniform; EDGE = false: xb[-145 .. 2047] are samples of this call
    // ---- stage d[blk0 - 144 .. blk0 + 2047]: a pair of samples per lane and pass, all loads of the workgroup issued before the first use
    sm_f32x4 v[SM_NPASS];
    sm_f32x2 pv[SM_NPASS];
#pragma unroll
    for (int ps = 0; ps < SM_NPASS; ps++) {
        const int j = 2 * tid + ps * 2 * SM_THREADS;
        const bool in = !EDGE || (blk0 + j >= 1 && blk0 + j + 2 <= n_total);      // pairs at the edges are fetched sample by sample below
        v[ps] = __builtin_nontemporal_load(reinterpret_cast<const sm_f32x4 *>(in ? xb + j : x));
        pv[ps] = __builtin_nontemporal_load(reinterpret_cast<const sm_f32x2 *>(in ? xb + j - 1 : x));
    }
    if (tid < SM_HALO / 2) {
        const int j = 2 * tid - SM_HALO;
        if (EDGE) sm_stage(lds, 2 * tid, sm_fetch(x, xh, n_total, blk0 + j - 1), sm_fetch(x, xh, n_total, blk0 + j), sm_fetch(x, xh, n_total, blk0 + j + 1));
        else sm_stage(lds, 2 * tid, xb[j - 1], xb[j], xb[j + 1]);
    }
    sm_bf16x8 TP[3], TS[2];
#pragma unroll
    for (int s = 0; s < 3; s++) TP[s] = __builtin_bit_cast(sm_bf16x8, frag[s * 64 + lane]);
#pragma unroll
    for (int s = 0; s < 2; s++) TS[s] = __builtin_bit_cast(sm_bf16x8, frag[(3 + 1 + s) * 64 + lane]);
#pragma unroll
    for (int ps = 0; ps < SM_NPASS; ps++) {
        const int j = 2 * tid + ps * 2 * SM_THREADS;
        float2 a = make_float2(pv[ps].x, pv[ps].y), b = make_float2(v[ps].x, v[ps].y), c = make_float2(v[ps].z, v[ps].w);
        if (EDGE && !(blk0 + j >= 1 && blk0 + j + 2 <= n_total)) {
            a = sm_fetch(x, xh, n_total, blk0 + j - 1); b = sm_fetch(x, xh, n_total, blk0 + j); c = sm_fetch(x, xh, n_total, blk0 + j + 1);
        }
        sm_stage(lds, SM_HALO + j, a, b, c);
    }
    __syncthreads();

    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t2 = 0; t2 < SM_TILE / 256 / (SM_THREADS / 64); t2++) {
        const int tt = wv + t2 * (SM_THREADS / 64);
        const long long o0 = blk0 + 256 * tt;
        if (EDGE && o0 >= n_total) break;
        const bool tail = EDGE && FUSED && o0 + 256 > n_total - 64;  // this tile holds some of the call's last 64 outputs
        sm_f32x4 P[2], S[2], S2[2];
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            // windows of block c + 16 tt: K step s of the band at plane index 64 + 16 (c + 2 s + 16 tt) + 8 g = fragment s + 1 below; the delayed
            // SOF correlation reads its K steps 1, 2 64 entries earlier (fragments 0, 1), the SOF correlation in place fragments 2, 3
            // parts smallest first, the three chains side by side (independent accumulators, four fragments live at a time)
            sm_f32x4 dP = {0.f, 0.f, 0.f, 0.f}, dS = {0.f, 0.f, 0.f, 0.f}, dT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 2; p >= 0; p--) {
                sm_bf16x8 W[4];                                     // fragment f at plane index 32 + 16 (c + 2 f + 16 tt) + 8 g
#pragma unroll
                for (int f = FUSED ? 0 : 1; f < 4; f++)
                    W[f] = __builtin_bit_cast(sm_bf16x8, *reinterpret_cast<const uint4 *>(lds + (2 * p + pl) * SM_PLANE + 32 + 16 * (c + 2 * f + 16 * tt) + 8 * g));
#pragma unroll
                for (int s = 0; s < 3; s++) {
                    dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s + 1], TP[s], dP, 0, 0, 0);
                    if (s < 2) dS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s + (FUSED ? 0 : 2)], TS[s], dS, 0, 0, 0);
                    if (s < 2 && tail) dT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[s + 2], TS[s], dT, 0, 0, 0);
                }
            }
            P[pl] = dP; S[pl] = dS; S2[pl] = dT;
        }
        const int ol = 256 * tt + 64 * g + c;                        // output of accumulator register r: blk0 + ol + 16 r
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long o = blk0 + ol + 16 * r;
            if (EDGE && o >= n_total) continue;
            if (FUSED) {
                float2 s = make_float2(S[0][r], S[1][r]);
                if (EDGE && o < 64) s = sofh[o];                    // cor_SOF of the 64 samples before this call
                const float2 p = make_float2(P[0][r], P[1][r]);
                const float sr = p.x + s.x, si = p.y + s.y, dr = s.x - p.x, di = s.y - p.y;
                const float a2s = fmaf(sr, sr, si * si), a2d = fmaf(dr, dr, di * di);
                (corr + blk0)[ol + 16 * r] = sqrtf(fmaxf(a2s, a2d));
                if (tail && o >= n_total - 64) sofh_out[o - (n_total - 64)] = make_float2(S2[0][r], S2[1][r]);
            } else {
                (cor_plsc + blk0)[ol + 16 * r] = make_float2(P[0][r], P[1][r]);
                (cor_sof + blk0)[ol + 16 * r] = make_float2(S[0][r], S[1][r]);
            }
        }
    }
}

// frag: [PLSC | SOF][K step][lane] band fragments (host-made, 6 KB, L2-resident)
// (26 KB of LDS leave six workgroups per CU; the register allocator is told so, or it settles for 104 registers = four)
template <bool FUSED>
__global__ void __launch_bounds__(SM_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8)))
sync_corr_mfma_kernel(const float2 *__restrict__ x, const float2 *__restrict__ xh, float2 *__restrict__ xh_out, const uint4 *__restrict__ frag, const float2 *__restrict__ sofh,
                      float2 *__restrict__ sofh_out, float *__restrict__ corr, float2 *__restrict__ cor_sof, float2 *__restrict__ cor_plsc, long long n_total)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[6 * SM_PLANE];
    const long long blk0 = (long long)blockIdx.x * SM_TILE;
    // the memory of the next call = the last 64 samples of (old memory ++ x); xh_out is not xh
    if (blockIdx.x == 0 && threadIdx.x >= SM_THREADS - SM_H) {
        const long long gi = n_total - SM_THREADS + threadIdx.x;
        xh_out[threadIdx.x - (SM_THREADS - SM_H)] = gi >= 0 ? x[gi] : xh[SM_H + gi];
    }
    if (blk0 >= SM_HALO + 1 && blk0 + SM_TILE + 2 <= n_total - 64) sm_body<FUSED, false>(lds, x, xh, frag, sofh, sofh_out, corr, cor_sof, cor_plsc, n_total, blk0);
    else sm_body<FUSED, true>(lds, x, xh, frag, sofh, sofh_out, corr, cor_sof, cor_plsc, n_total, blk0);
}

static inline uint16_t sm_bf16_of(float v) { uint32_t u; std::memcpy(&u, &v, 4); return (uint16_t)(u >> 16); }     // +-1 and 0: exact

// [PLSC | SOF][K step][lane][8]: B operand of the band A[i][k] = brev81[k - i] (taps reversed, right-aligned in 81 entries), lane (c, g) holds
// rows k = 32 s + 8 g + j of column c
std::vector<uint16_t> sync_mfma_frag(const float *sof25, const float *plsc64)
{
    std::vector<uint16_t> out((size_t)2 * 3 * 64 * 8, 0);
    for (int q = 0; q < 2; q++) {
        const int T = q == 0 ? 64 : 25;
        const float *b = q == 0 ? plsc64 : sof25;
        for (int s = 0; s < 3; s++)
            for (int l = 0; l < 64; l++)
                for (int j = 0; j < 8; j++) {
                    const int i = l & 15, k = 32 * s + 8 * (l >> 4) + j, m = 80 - (k - i);       // tap that pairs d[o - m] with output o
                    out[(((size_t)q * 3 + s) * 64 + l) * 8 + j] = (m >= 0 && m < T) ? sm_bf16_of(b[m]) : 0;
                }
    }
    return out;
}

bool sync_mfma_usable(const float *x, const void *frag) { return frag && (reinterpret_cast<uintptr_t>(x) & 15) == 0; }

// two-task form: both correlations to their sockets
hipError_t sync_corr_mfma_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, float *cor_sof, float *cor_plsc, long long n_total, hipStream_t s)
{
    const unsigned grid = (unsigned)((n_total + SM_TILE - 1) / SM_TILE);
    hipLaunchKernelGGL(sync_corr_mfma_kernel<false>, dim3(grid), dim3(SM_THREADS), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(xh_in),
                       reinterpret_cast<float2 *>(xh_out), reinterpret_cast<const uint4 *>(frag), nullptr, nullptr, nullptr, reinterpret_cast<float2 *>(cor_sof), reinterpret_cast<float2 *>(cor_plsc), n_total);
    return hipGetLastError();
}

// one-task form: the instantaneous metric only
hipError_t sync_corr_m_mfma_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, const float *sofh_in, float *sofh_out, float *corr,
                                   long long n_total, hipStream_t s)
{
    const unsigned grid = (unsigned)((n_total + SM_TILE - 1) / SM_TILE);
    hipLaunchKernelGGL(sync_corr_mfma_kernel<true>, dim3(grid), dim3(SM_THREADS), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(xh_in),
                       reinterpret_cast<float2 *>(xh_out), reinterpret_cast<const uint4 *>(frag), reinterpret_cast<const float2 *>(sofh_in), reinterpret_cast<float2 *>(sofh_out), corr, nullptr, nullptr,
                       n_total);
    return hipGetLastError();
}

}  // namespace dvbs2
