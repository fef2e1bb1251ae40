// a5 -- SRRC matched filter: complex input, real taps, streaming FIR for gfx950.
//
// Replaces Filter_FIR_ccr<R>::_filter (+ step())
// (/root/reference src/common/Module/Filter/Filter_FIR/Filter_FIR_ccr.cpp:68-142,
//  Filter_FIR_ccr.hpp:39-52):   y[i] = sum_k brev[k] * x[i - (T-1) + k],  brev[k] = b[T-1-k]
// where x[<0] is the tail the previous call left behind (Filter_FIR_ccr.cpp:80-83).  The F
// frames of a socket are consecutive in time, so the whole batch is ONE stream of
// F * n_cplx samples and the reference's per-frame scalar head (`step()`) is just the same sum
// reading the previous frame: no special case is needed (overlap-save, H7).
//
// 324 flop per complex sample against 16 B: AI ~ 20 flop/B, i.e. at the fp32 ridge of the chip,
// so the kernel is built compute-first: each lane produces 8 consecutive outputs from an
// 88-sample window held in a sliding register file (16 FMAs per LDS read), the tile is staged
// once through LDS with a +1-per-8 pad that makes the stride-8 window reads conflict-free, and
// the taps are wave-uniform scalars.  fp32 FMA like the reference's mipp::fmadd; the f32 MFMA
// runs at the same rate as the vector FMA on gfx950 and a Toeplitz recast would waste flops on
// the structural zeros; the matrix-core form that does pay -- bf16 x 3 split operands -- is k_fir_mfma.hip, the default
// for T <= 81, and this kernel serves longer filters, unaligned sockets and DVBS2HIP_FIR=valu.
#include "dvbs2hip_internal.h"
#include <cstdlib>

namespace dvbs2 {

constexpr int FIR_THREADS = 256;
constexpr int FIR_R = 8;                          // outputs per lane
constexpr int FIR_TILE = FIR_THREADS * FIR_R;     // outputs per workgroup
constexpr int FIR_TMAX = 257;

__host__ __device__ __forceinline__ int fir_pad(int s) { return s + (s >> 3); }

template <int TS>   // TS > 0: compile-time tap count (fully unrolled); TS == 0: run-time T
__global__ void __launch_bounds__(FIR_THREADS)
fir_ccr_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, const float2 *__restrict__ hist_in,
               const float *__restrict__ taps_rev, int Trt, long long n_total)
{
    extern __shared__ float2 tile[];             // fir_pad(FIR_TILE + T - 1) samples
    const int T = TS > 0 ? TS : Trt;
    const int H = T - 1;
    const int tid = threadIdx.x;
    const long long blk0 = (long long)blockIdx.x * FIR_TILE;
    const int n_in = FIR_TILE + H;
    for (int s = tid; s < n_in; s += FIR_THREADS) {
        const long long gi = blk0 - H + s;
        float2 v = make_float2(0.f, 0.f);
        if (gi < 0) v = hist_in[H + gi];
        else if (gi < n_total) v = x[gi];
        tile[fir_pad(s)] = v;
    }
    __syncthreads();
    const int s0 = tid * FIR_R;                   // window start inside the tile
    float2 acc[FIR_R], w[FIR_R];
    // s0 is a multiple of 8, so fir_pad(s0 + j) = 9 tid + j + (j >> 3): one per-lane base, every further index a constant
    // that goes into the immediate offset of the ds_read (no address arithmetic per tap)
    const float2 *tp = tile + 9 * tid;
    static_assert(FIR_R == 8, "the padded index is split on a window of 8");
#pragma unroll
    for (int r = 0; r < FIR_R; r++) { acc[r] = make_float2(0.f, 0.f); w[r] = tp[r]; }
    if (TS > 0) {
#pragma unroll
        for (int k = 0; k < (TS > 0 ? TS : 1); k++) {
            const float b = taps_rev[k];
#pragma unroll
            for (int r = 0; r < FIR_R; r++) { acc[r].x = fmaf(b, w[r].x, acc[r].x); acc[r].y = fmaf(b, w[r].y, acc[r].y); }
#pragma unroll
            for (int r = 0; r + 1 < FIR_R; r++) w[r] = w[r + 1];
            if (k + 1 < TS) w[FIR_R - 1] = tp[(k + FIR_R) + ((k + FIR_R) >> 3)];
        }
    } else {
        for (int k = 0; k < T; k++) {
            const float b = taps_rev[k];
#pragma unroll
            for (int r = 0; r < FIR_R; r++) { acc[r].x = fmaf(b, w[r].x, acc[r].x); acc[r].y = fmaf(b, w[r].y, acc[r].y); }
#pragma unroll
            for (int r = 0; r + 1 < FIR_R; r++) w[r] = w[r + 1];
            if (k + 1 < T) w[FIR_R - 1] = tile[fir_pad(s0 + k + FIR_R)];
        }
    }
    const long long o0 = blk0 + s0;
    if (o0 + FIR_R <= n_total) {
        float4 *dst = reinterpret_cast<float4 *>(y + o0);
#pragma unroll
        for (int r = 0; r < FIR_R; r += 2) dst[r / 2] = make_float4(acc[r].x, acc[r].y, acc[r + 1].x, acc[r + 1].y);
    } else {
#pragma unroll
        for (int r = 0; r < FIR_R; r++) if (o0 + r < n_total) y[o0 + r] = acc[r];
    }
}

// new history = last H samples of (old history ++ x)
__global__ void fir_hist_kernel(const float2 *x, const float2 *hist_in, float2 *hist_out, int H, long long n_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    const long long gi = n_total - H + i;
    hist_out[i] = gi >= 0 ? x[gi] : hist_in[H + gi];
}

// afrag != nullptr: the matrix-core form (k_fir_mfma.hip) when it applies (T <= 81, 16-byte aligned sockets);
// dvbs2hip_set_filter_kernel(h, DVBS2HIP_FIR_VALU) (afrag == nullptr here) or DVBS2HIP_FIR=valu keep the vector kernel
hipError_t fir_launch(const float *x, float *y, const float *hist_in, float *hist_out, const float *taps_rev, const uint16_t *afrag,
                      int T, long long n_total, hipStream_t s)
{
    if (T < 1 || T > FIR_TMAX) return hipErrorInvalidValue;
    const int H = T - 1;
    static const bool force_valu = [] { const char *e = getenv("DVBS2HIP_FIR"); return e && e[0] == 'v'; }();
    if (afrag && !force_valu && fir_mfma_usable(x, y, T, n_total))
        return fir_mfma_launch(x, y, hist_in, hist_out, afrag, T, n_total, s);
    const size_t lds = sizeof(float2) * (size_t)(fir_pad(FIR_TILE + H) + 1);
    const unsigned grid = (unsigned)((n_total + FIR_TILE - 1) / FIR_TILE);
    const float2 *x2 = reinterpret_cast<const float2 *>(x);
    float2 *y2 = reinterpret_cast<float2 *>(y);
    const float2 *h2 = reinterpret_cast<const float2 *>(hist_in);
    if (T == 81) hipLaunchKernelGGL(fir_ccr_kernel<81>, dim3(grid), dim3(FIR_THREADS), lds, s, x2, y2, h2, taps_rev, T, n_total);
    else         hipLaunchKernelGGL(fir_ccr_kernel<0>, dim3(grid), dim3(FIR_THREADS), lds, s, x2, y2, h2, taps_rev, T, n_total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (H > 0)
        hipLaunchKernelGGL(fir_hist_kernel, dim3((H + 63) / 64), dim3(64), 0, s, x2, h2,
                           reinterpret_cast<float2 *>(hist_out), H, n_total);
    return hipGetLastError();
}


// ---------------------------------------------------------------- N2: TX shaping filter (polyphase up-sampling FIR)
// Replaces Filter_UPRRC_ccr_naive / Filter_UPFIR_ccr_naive::_filter
// (/root/reference src/common/Module/Filter/Filter_UPFIR/Filter_UPFIR_ccr_naive.cpp:52-66): a bank of
// `osf` FIRs, branch f holds taps H[f], H[f+osf], ..; output sample i*osf + f = branch f at input i:
//     y[i*osf + f] = sum_m H[f + m*osf] * x[i - m]
// One lane per output sample, inputs staged in LDS; x[<0] = tail of the previous call.
constexpr int UP_THREADS = 256;
__global__ void __launch_bounds__(UP_THREADS)
upfir_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, const float2 *__restrict__ hist_in, const float *__restrict__ taps,
             int T, int osf, int Hin, long long n_in)
{
    extern __shared__ float2 xt[];                 // (UP_THREADS / osf + Hin) input samples, then the taps
    const int per_blk = UP_THREADS / osf;          // input samples per workgroup
    float *tp = reinterpret_cast<float *>(xt + per_blk + Hin);
    const long long in0 = (long long)blockIdx.x * per_blk;
    for (int s = threadIdx.x; s < per_blk + Hin; s += UP_THREADS) {
        const long long gi = in0 - Hin + s;
        float2 v = make_float2(0.f, 0.f);
        if (gi < 0) v = hist_in[Hin + gi]; else if (gi < n_in) v = x[gi];
        xt[s] = v;
    }
    for (int s = threadIdx.x; s < T; s += UP_THREADS) tp[s] = taps[s];
    __syncthreads();
    const int li = threadIdx.x / osf, f = threadIdx.x - li * osf;
    const long long i = in0 + li;
    if (li >= per_blk || i >= n_in) return;
    float2 acc = make_float2(0.f, 0.f);
    for (int m = 0, j = f; j < T; m++, j += osf) {
        const float2 v = xt[Hin + li - m];
        acc.x = fmaf(tp[j], v.x, acc.x); acc.y = fmaf(tp[j], v.y, acc.y);
    }
    y[i * osf + f] = acc;
}

// afrag2 != nullptr: the matrix-core form (k_fir_mfma.hip) when it applies (osf = 2, branches of at most 81 taps, aligned sockets)
hipError_t upfir_launch(const float *x, float *y, const float *hist_in, float *hist_out, const float *taps, const uint16_t *afrag2, int T, int osf,
                        long long n_in, hipStream_t s)
{
    if (T < 1 || T > FIR_TMAX || osf < 1 || osf > 16) return hipErrorInvalidValue;
    if (afrag2 && upfir_mfma_usable(x, y, T, osf, n_in)) return upfir_mfma_launch(x, y, hist_in, hist_out, afrag2, T, n_in, s);
    const int Hin = (T - 1) / osf;                 // input samples of memory
    const int per_blk = UP_THREADS / osf;
    const size_t lds = sizeof(float2) * (size_t)(per_blk + Hin) + sizeof(float) * (size_t)T;
    const unsigned grid = (unsigned)((n_in + per_blk - 1) / per_blk);
    hipLaunchKernelGGL(upfir_kernel, dim3(grid), dim3(UP_THREADS), lds, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<float2 *>(y),
                       reinterpret_cast<const float2 *>(hist_in), taps, T, osf, Hin, n_in);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (Hin > 0)
        hipLaunchKernelGGL(fir_hist_kernel, dim3((Hin + 63) / 64), dim3(64), 0, s, reinterpret_cast<const float2 *>(x),
                           reinterpret_cast<const float2 *>(hist_in), reinterpret_cast<float2 *>(hist_out), Hin, n_in);
    return hipGetLastError();
}

// ---------------------------------------------------------------- perfect-timing extraction: y[i] = x[offset + i * osf]
// (what Synchronizer_timing_perfect does once the delay is known: DVBS2.cpp:558-570)
__global__ void decimate_kernel(const float2 *x, float2 *y, long long n_out, int osf, long long offset, long long n_in)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const long long s = offset + i * osf;
    y[i] = (s >= 0 && s < n_in) ? x[s] : make_float2(0.f, 0.f);
}
hipError_t decimate_launch(const float *x, float *y, long long n_out, int osf, long long offset, long long n_in, hipStream_t s)
{
    hipLaunchKernelGGL(decimate_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<float2 *>(y), n_out, osf, offset, n_in);
    return hipGetLastError();
}

}  // namespace dvbs2
