// The two element-wise stages at the front of the reference's RX graph that are block-wise tasks: the gain stage and the frequency shift of the coarse synchronizer.
//
// Multiplier_AGC_cc_naive::_imultiply (src/common/Module/Multiplier/Sequence/Multiplier_AGC_cc_naive.cpp:22-46): every frame is brought to a given energy --
//     std = sqrt(N sum|x|^2 - (sum re)^2 - (sum im)^2) / N / sqrt(output_energy),   z = x / std
// (the frame's standard deviation about its mean).  The reference's RX graph runs it twice: `front_agc` on the received samples (2 pl_frame osf values per frame,
// energy 1 / osf: RX/main_sched.cpp:74,197; DVBS2.cpp:660-664) and `mult_agc` on the symbols behind the timing synchronizer (energy 1: main_sched.cpp:75,205; DVBS2.cpp:653-657).
// A block-wise task: one workgroup per frame, the frame read twice (sums, then scale: the second read comes out of L2), 8 + 8 bytes per complex sample at the boundary.
// The three sums are taken in double precision over a fixed tree, so a frame's gain does not depend on the batch or the launch; the reference adds its floats in order,
// which rounds differently: the parity bar is relative (tests/test_front_gpu.py: 2e-6 against the sums in double, 1e-4 against the oracle's float order).
#include "dvbs2hip_internal.h"

namespace dvbs2 {

constexpr int AGC_THREADS = 256;

__global__ void __launch_bounds__(AGC_THREADS)
agc_kernel(const float2 *__restrict__ x, float2 *__restrict__ z, int n_cplx, float output_energy)
{
    const float2 *xf = x + (size_t)blockIdx.x * n_cplx;
    float2 *zf = z + (size_t)blockIdx.x * n_cplx;
    double s2 = 0.0, sr = 0.0, si = 0.0;
    for (int i = threadIdx.x; i < n_cplx; i += AGC_THREADS) {
        const float2 v = xf[i];
        s2 += (double)v.x * v.x + (double)v.y * v.y;
        sr += v.x;
        si += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o); sr += __shfl_xor(sr, o); si += __shfl_xor(si, o); }
    __shared__ double part[3][AGC_THREADS / 64];
    __shared__ float s_std;
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s2; part[1][threadIdx.x >> 6] = sr; part[2][threadIdx.x >> 6] = si; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]), b = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]), c = (part[2][0] + part[2][1]) + (part[2][2] + part[2][3]);
        // (an all-equal frame gives 0 -> z = x / 0 = +-inf or NaN, as in the reference)
        s_std = (float)(sqrt(fmax(a * (double)n_cplx - b * b - c * c, 0.0)) / (double)n_cplx) / sqrtf(output_energy);
    }
    __syncthreads();
    const float sd = s_std;
    for (int i = threadIdx.x; i < n_cplx; i += AGC_THREADS) {
        const float2 v = xf[i];
        zf[i] = make_float2(v.x / sd, v.y / sd);
    }
}

// Frames of up to 20480 complex samples (every short-frame MODCOD at osf 2, the normal frames' symbols up to 16APSK): the frame stays in registers between the sums and the
// division -- 16 workgroup-wide loads of 8 bytes per lane at most, one read and one write of the frame instead of two reads (0.341 -> 0.229 ms per 4096 QPSK-S frames of samples, docs/kernels.md).
// Which kernel runs depends on the frame length alone, so a frame's gain still does not depend on the batch.
constexpr int AGC_WIDE = 1024;
typedef float agc_f2 __attribute__((ext_vector_type(2)));
template <int R>
__global__ void __launch_bounds__(AGC_WIDE)
agc_reg_kernel(const float2 *__restrict__ x, float2 *__restrict__ z, int n_cplx, float output_energy)
{
    const float2 *xf = x + (size_t)blockIdx.x * n_cplx;
    float2 *zf = z + (size_t)blockIdx.x * n_cplx;
    float2 v[R];
    double s2 = 0.0, sr = 0.0, si = 0.0;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = k * AGC_WIDE + (int)threadIdx.x;
        if (i < n_cplx) { const agc_f2 q = __builtin_nontemporal_load(reinterpret_cast<const agc_f2 *>(xf) + i); v[k] = make_float2(q.x, q.y); }
        else v[k] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < R; k++) { s2 += (double)v[k].x * v[k].x + (double)v[k].y * v[k].y; sr += v[k].x; si += v[k].y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o); sr += __shfl_xor(sr, o); si += __shfl_xor(si, o); }
    __shared__ double part[3][AGC_WIDE / 64];
    __shared__ float s_std;
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s2; part[1][threadIdx.x >> 6] = sr; part[2][threadIdx.x >> 6] = si; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int w = 0; w < AGC_WIDE / 64; w++) { a += part[0][w]; b += part[1][w]; c += part[2][w]; }
        s_std = (float)(sqrt(fmax(a * (double)n_cplx - b * b - c * c, 0.0)) / (double)n_cplx) / sqrtf(output_energy);
    }
    __syncthreads();
    const float sd = s_std;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const int i = k * AGC_WIDE + (int)threadIdx.x;
        if (i < n_cplx) __builtin_nontemporal_store(agc_f2{v[k].x / sd, v[k].y / sd}, reinterpret_cast<agc_f2 *>(zf) + i);
    }
}

hipError_t agc_launch(const float *X, float *Z, int n_cplx, float output_energy, int F, hipStream_t s)
{
    const float2 *x = reinterpret_cast<const float2 *>(X);
    float2 *z = reinterpret_cast<float2 *>(Z);
    const int r = (n_cplx + AGC_WIDE - 1) / AGC_WIDE;
    if (r <= 4) hipLaunchKernelGGL(agc_reg_kernel<4>, dim3(F), dim3(AGC_WIDE), 0, s, x, z, n_cplx, output_energy);
    else if (r <= 9) hipLaunchKernelGGL(agc_reg_kernel<9>, dim3(F), dim3(AGC_WIDE), 0, s, x, z, n_cplx, output_energy);
    else if (r <= 17) hipLaunchKernelGGL(agc_reg_kernel<17>, dim3(F), dim3(AGC_WIDE), 0, s, x, z, n_cplx, output_energy);
    else if (r <= 20) hipLaunchKernelGGL(agc_reg_kernel<20>, dim3(F), dim3(AGC_WIDE), 0, s, x, z, n_cplx, output_energy);
    else hipLaunchKernelGGL(agc_kernel, dim3(F), dim3(AGC_THREADS), 0, s, x, z, n_cplx, output_energy);
    return hipGetLastError();
}

// Synchronizer_freq_coarse_DVBS2_aib::_synchronize in the transmission phase = Multiplier_sine_ccc_naive::imultiply with the frequency the loop has settled on
// (Synchronizer_freq_coarse_DVBS2_aib.cpp:43-50; Multiplier_sine_ccc_naive.cpp:69-77: phase = omega n in the module's float type, z = x (cos phase + j sin phase), n counts
// the samples of the stream and starts over after 999999 -- nu is kept to six decimals, set_nu :44-51, so that omega 1e6 is a whole number of turns).  A block-wise task: the
// phase of sample i is a closed form of the stream position, nothing is carried from sample to sample.
__global__ void __launch_bounds__(256)
nco_kernel(const float2 *__restrict__ x, float2 *__restrict__ z, float omega, uint32_t n0, long long total, float *FRQ, float *PHS, float frq, int F)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < F) { if (FRQ) FRQ[i] = frq; if (PHS) PHS[i] = 0.f; }      // the module's estimated_freq / estimated_phase per frame (Synchronizer_freq_coarse.hxx:40-47): constant while its loop does not run
    if (i >= total) return;
    const float n = (float)(uint32_t)(((unsigned long long)n0 + (unsigned long long)i) % 1000000ull);
    const float phase = omega * n;
    float sn, cs;
    sincosf(phase, &sn, &cs);
    const float2 v = x[i];
    z[i] = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
}

hipError_t nco_launch(const float *X, float *Z, float omega, uint32_t n0, long long total, float *FRQ, float *PHS, float frq, int F, hipStream_t s)
{
    hipLaunchKernelGGL(nco_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2 *>(X), reinterpret_cast<float2 *>(Z), omega, n0, total, FRQ, PHS, frq, F);
    return hipGetLastError();
}

}  // namespace dvbs2
