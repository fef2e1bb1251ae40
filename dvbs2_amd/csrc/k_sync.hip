// N4 -- frame synchronizer, the data-parallel form of Synchronizer_frame_DVBS2_fast
// (/root/reference src/common/Module/Synchronizer/Synchronizer_frame/Synchronizer_frame_DVBS2_fast.cpp):
//   _synchronize1 (:132-150)  differential signal d[i] = x[i-1] conj(x[i]) and the two correlators
//                             corr_SOF (25 real taps) / corr_PLSC (64 real taps), both Filter_FIR_ccr
//   _synchronize2 (:222-299)  cor_SOF delayed by 64, |sum|^2 / |difference|^2 of the two correlations,
//                             alpha-average per position across frames, arg max -> delay, and the
//                             variable output delay (Variable_delay_cc_naive::_filter)
// A call carries F PL frames that are consecutive in time = F calls of the reference with n_frames = 1.
// Everything is a stream FIR or an element-wise map except three small recurrences across frames
// (the average per position, the arg max per frame, the delay line), which run frame after frame on
// device-resident state: no host round trip between the stages.
#include "dvbs2hip_internal.h"
#include <cstring>
#include <type_traits>

namespace dvbs2 {

// conj_SOF / conj_PLSC of Synchronizer_frame_DVBS2_fast.hpp:19-33 (compile-time: the taps are +-1 and, for the PLSC, every
// other one is 0 -- fma(0, v, acc) is acc, so leaving those out changes nothing)
constexpr float K_CONJ_SOF[25] = {1, -1, -1, 1, -1, 1, 1, -1, 1, 1, -1, -1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, 1, 1, 1};
constexpr float K_CONJ_PLSC[64] = {1, 0, 1, 0, -1, 0, -1, 0, 1, 0, -1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1, 0, 1, 0, -1, 0, -1, 0, 1, 0, 1, 0,
                                   1, 0, 1, 0, -1, 0, -1, 0, -1, 0, -1, 0, -1, 0, 1, 0, 1, 0, -1, 0, 1, 0, 1, 0, 1, 0, -1, 0, -1, 0, 1, 0};

constexpr int SY_THREADS = 256;
constexpr int SY_R = 4;                       // outputs per lane
constexpr int SY_T = SY_THREADS * SY_R;       // outputs per workgroup
constexpr int SY_H = 64;                      // samples of x a block needs before its first output (63 of d, one more of x)
__device__ __forceinline__ int sy_pad(int i) { return i + (i >> 2); }     // one element of padding per 4: the lanes' stride-4 reads hit 32 different banks

// ---- _synchronize1: x -> cor_SOF, cor_PLSC.  xh = the last 64 samples of the stream so far
// (initially zeros with (1, 0) last: reg_channel, :19; the correlators start from empty memories).
// Each lane forms SY_R consecutive outputs from one pass over the 63 + SY_R differential samples they share (a value is read
// from LDS once and used by up to SY_R x 2 taps; one output per lane read 89 values per output and was LDS-bound), every
// output accumulating oldest sample first like Filter_FIR_ccr.cpp:68-142.
__global__ void __launch_bounds__(SY_THREADS)
sync_corr_kernel(const float2 *__restrict__ x, const float2 *__restrict__ xh, float2 *__restrict__ cor_sof, float2 *__restrict__ cor_plsc,
                 long long n_total)
{
    __shared__ float2 xs[SY_T + SY_H];                               // xs[k] = x[blk0 - 64 + k]
    __shared__ float2 ds[SY_T + SY_H + (SY_T + SY_H) / 4 + 4];       // ds[pad(k)] = d[blk0 - 63 + k] = x[blk0 - 64 + k] conj(x[blk0 - 63 + k])
    const long long blk0 = (long long)blockIdx.x * SY_T;
    for (int k = threadIdx.x; k < SY_T + SY_H; k += SY_THREADS) {
        const long long g = blk0 - SY_H + k;
        float2 v = make_float2(0.f, 0.f);
        if (g < 0) v = xh[SY_H + g]; else if (g < n_total) v = x[g];
        xs[k] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SY_T + SY_H - 1; k += SY_THREADS) {
        const float2 a = xs[k], b = xs[k + 1];                     // :138-142 (a = previous sample)
        ds[sy_pad(k)] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
    }
    __syncthreads();
    const int l0 = threadIdx.x * SY_R;                             // first output of the lane inside the block
    const long long i0 = blk0 + l0;
    if (i0 >= n_total) return;
    // output l0 + r = sum_m b[m] d[i - m]; d[i - m] sits at ds index (l0 + r) + 63 - m
    float2 ap[SY_R], as[SY_R];
#pragma unroll
    for (int r = 0; r < SY_R; r++) { ap[r] = make_float2(0.f, 0.f); as[r] = make_float2(0.f, 0.f); }
#pragma unroll
    for (int jj = 0; jj < 64 + SY_R - 1; jj++) {                   // ds index l0 + jj, oldest first
        const float2 v = ds[sy_pad(l0 + jj)];
#pragma unroll
        for (int r = 0; r < SY_R; r++) {
            const int m = 63 + r - jj;                             // tap that pairs this sample with output r
            if (m >= 0 && m < 64 && K_CONJ_PLSC[m < 0 ? 0 : m > 63 ? 63 : m] != 0.f) {
                const float b = K_CONJ_PLSC[m < 0 ? 0 : m > 63 ? 63 : m];
                ap[r].x = fmaf(b, v.x, ap[r].x); ap[r].y = fmaf(b, v.y, ap[r].y);
            }
            if (m >= 0 && m < 25) {
                const float b = K_CONJ_SOF[m < 0 ? 0 : m > 24 ? 24 : m];
                as[r].x = fmaf(b, v.x, as[r].x); as[r].y = fmaf(b, v.y, as[r].y);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < SY_R; r++) if (i0 + r < n_total) { cor_plsc[i0 + r] = ap[r]; cor_sof[i0 + r] = as[r]; }
}

// ---- _synchronize (the one-task form, :46-128): the two correlations are not sockets there, so they never leave the chip.  Same
// arithmetic as sync_corr_kernel + sync_m_kernel, output for output: a block forms cor_PLSC of its SY_T samples and cor_SOF of
// those and of the 64 samples before them (the metric pairs cor_PLSC[g] with cor_SOF[g - 64], :236), keeps cor_SOF in LDS and
// writes only the instantaneous metric m[g]: 8 B read + 4 B written per sample instead of 24 + 20.  The first block takes the
// delayed cor_SOF from the handle's history (sofh), the last 64 cor_SOF values of the stream go to sofh_out.
typedef float sy_f2 __attribute__((ext_vector_type(2)));
constexpr int SY_HX = 96;                     // samples of x before the block: 64 (SOF halo) + 24 (its taps) + 1 (differential), rounded up
__global__ void __launch_bounds__(SY_THREADS)
sync_corr_m_kernel(const float2 *__restrict__ x, const float2 *__restrict__ xh, const float2 *__restrict__ sofh, float2 *__restrict__ sofh_out,
                   float *__restrict__ corr, long long n_total)
{
    __shared__ float2 xs[SY_T + SY_HX];                              // xs[k] = x[blk0 - 96 + k]
    __shared__ float2 ds[SY_T + SY_HX + (SY_T + SY_HX) / 4 + 4];     // ds[pad(k)] = d[blk0 - 95 + k]
    __shared__ float2 sof_s[SY_T + 64];                              // sof_s[k] = cor_SOF[blk0 - 64 + k]
    const long long blk0 = (long long)blockIdx.x * SY_T;
    for (int k = threadIdx.x; k < SY_T + SY_HX; k += SY_THREADS) {
        const long long g = blk0 - SY_HX + k;
        float2 v = make_float2(0.f, 0.f);
        if (g < 0) { if (g >= -SY_H) v = xh[SY_H + g]; } else if (g < n_total) v = x[g];
        xs[k] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SY_T + SY_HX - 1; k += SY_THREADS) {
        const float2 a = xs[k], b = xs[k + 1];
        ds[sy_pad(k)] = make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
    }
    if (blk0 == 0 && threadIdx.x < 64) sof_s[threadIdx.x] = sofh[threadIdx.x];       // cor_SOF of the 64 samples before this call
    __syncthreads();
    const int l0 = threadIdx.x * SY_R;
    const long long i0 = blk0 + l0;
    // d[i - m] of output l (sample blk0 + l) sits at ds index l + 95 - m
    // the taps are +-1 (or 0), and fma(+-1, v, acc) = acc +- v exactly: packed adds of (re, im) pairs (v_pk_add_f32), half the vector
    // instructions of sync_corr_kernel's scalar fmas for the same bits
    sy_f2 ap[SY_R], as[SY_R];
#pragma unroll
    for (int r = 0; r < SY_R; r++) { ap[r] = sy_f2{0.f, 0.f}; as[r] = sy_f2{0.f, 0.f}; }
#pragma unroll
    for (int jj = 0; jj < 64 + SY_R - 1; jj++) {                   // ds index l0 + 32 + jj, oldest first: exactly sync_corr_kernel's sums
        const float2 vv = ds[sy_pad(l0 + 32 + jj)];
        const sy_f2 v = sy_f2{vv.x, vv.y};
#pragma unroll
        for (int r = 0; r < SY_R; r++) {
            const int m = 63 + r - jj;
            if (m >= 0 && m < 64 && K_CONJ_PLSC[m < 0 ? 0 : m > 63 ? 63 : m] != 0.f) {
                if (K_CONJ_PLSC[m < 0 ? 0 : m > 63 ? 63 : m] > 0.f) ap[r] = ap[r] + v; else ap[r] = ap[r] - v;
            }
            if (m >= 0 && m < 25) {
                if (K_CONJ_SOF[m < 0 ? 0 : m > 24 ? 24 : m] > 0.f) as[r] = as[r] + v; else as[r] = as[r] - v;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < SY_R; r++) sof_s[64 + l0 + r] = make_float2(as[r].x, as[r].y);
    if (blk0 > 0 && threadIdx.x < 64 / SY_R) {
        // cor_SOF of the 64 samples before the block (outputs l = -64 + 4 tid + r): 25 taps, oldest first
        const int h0 = threadIdx.x * SY_R;                          // sof_s index
        sy_f2 hs[SY_R];
#pragma unroll
        for (int r = 0; r < SY_R; r++) hs[r] = sy_f2{0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 25 + SY_R - 1; jj++) {                // ds index (h0 - 64) + 95 - 24 + jj = h0 + 7 + jj
            const float2 vv = ds[sy_pad(h0 + 7 + jj)];
            const sy_f2 v = sy_f2{vv.x, vv.y};
#pragma unroll
            for (int r = 0; r < SY_R; r++) {
                const int m = 24 + r - jj;
                if (m >= 0 && m < 25) { if (K_CONJ_SOF[m < 0 ? 0 : m > 24 ? 24 : m] > 0.f) hs[r] = hs[r] + v; else hs[r] = hs[r] - v; }
            }
        }
#pragma unroll
        for (int r = 0; r < SY_R; r++) sof_s[h0 + r] = make_float2(hs[r].x, hs[r].y);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SY_R; r++) {
        const long long g = i0 + r;
        if (g >= n_total) break;
        const float2 s = sof_s[l0 + r], p = make_float2(ap[r].x, ap[r].y);      // cor_SOF[g - 64], cor_PLSC[g]
        const float sr = p.x + s.x, si = p.y + s.y, dr = s.x - p.x, di = s.y - p.y;
        const float a2s = fmaf(sr, sr, si * si), a2d = fmaf(dr, dr, di * di);
        corr[g] = sqrtf(fmaxf(a2s, a2d));
        if (g >= n_total - 64) sofh_out[g - (n_total - 64)] = make_float2(as[r].x, as[r].y);
    }
}

// new history = last H samples of (old history ++ x)
__global__ void sync_hist_kernel(const float2 *x, const float2 *hist_in, float2 *hist_out, int H, long long n_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    const long long gi = n_total - H + i;
    hist_out[i] = gi >= 0 ? x[gi] : hist_in[H + gi];
}

// ---- _synchronize2, first half (:236-267, :278-285).  The instantaneous metric m[f][i] = max(|plsc + sof|, |sof - plsc|) of
// every sample is independent work (sync_m_kernel, one lane per sample of the batch); the average over frames
// corr_vec[i] = alpha corr_vec[i] + (1 - alpha) m[f][i] is a recurrence across frames and parallel across positions only
// (sync_metric_argmax_kernel below); cv = corr_vec (carried between calls).
__global__ void sync_m_kernel(const float2 *__restrict__ cor_sof, const float2 *__restrict__ sofh, const float2 *__restrict__ cor_plsc,
                              float *__restrict__ corr, long long n_total)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_total) return;
    const float2 s = g >= 64 ? cor_sof[g - 64] : sofh[g];            // SOF_PLSC_delay: 64 samples (:24, :236)
    const float2 p = cor_plsc[g];
    const float sr = p.x + s.x, si = p.y + s.y, dr = s.x - p.x, di = s.y - p.y;
    const float a2s = fmaf(sr, sr, si * si), a2d = fmaf(dr, dr, di * di);
    corr[g] = sqrtf(fmaxf(a2s, a2d));
}

// four wave-wide maxima side by side (the row_shr / row_bcast ladder of gfx9, the four registers interleaved so that no step waits for the
// data-parallel-primitive hazard of the one before): lane 63 of each register ends up with the wave's maximum
__device__ __forceinline__ void sy_wave_umax4(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
#define SY_DPP4(ctl) "v_max_u32_dpp %0, %0, %0 " ctl "\n\tv_max_u32_dpp %1, %1, %1 " ctl "\n\tv_max_u32_dpp %2, %2, %2 " ctl "\n\tv_max_u32_dpp %3, %3, %3 " ctl "\n\t"
    asm volatile("s_nop 1\n\t"
                 SY_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf") SY_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf")
                 SY_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf") SY_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf")
                 SY_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf") SY_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef SY_DPP4
}


// ---- the average over frames and the arg max of every frame (:269-285).
// One workgroup per 64 positions, four waves on the CU's four SIMDs.  The time of this stage is (frames) x (instructions a wave issues per frame):
// a position's chain runs over all F frames of the call and there are only n / 64 waves' worth of positions, so the chain is kept as short as
// it can be and the arg max work is taken off it.  Wave 0 runs the recurrence (the loads of a chunk of 24 frames in flight while the previous
// chunk goes through the dependent multiply-adds; same operations in the same order as the reference) and leaves the averaged values of a
// chunk in LDS; waves 1-3 take the wave-wide maxima of eight frames each (DPP ladder, four registers interleaved) while wave 0 is already on
// the next chunk -- one LDS-only barrier per chunk, two chunk buffers.  The averaged metric never goes to memory (it has no other reader).
// A frame's result is the 64-bit key (value bits << 32) | ~index, whose maximum over the workgroups is the largest value and, among equal
// values, the smallest index (= the reference's first maximum, :269-276; the metric is >= 0, so its bit pattern orders like the value).
// Every workgroup leaves its key of every frame in keys[frame][workgroup] (plain stores: agent-scope atomics on one word per frame from 521
// workgroups were what this kernel waited for); sync_finalize_kernel takes their maximum.  A frame without a value above 0 gets key 0.
constexpr int SYM_UF = 24;
#ifndef SYNC_ARGMAX10      // 1: the ten-wave form of the average / arg max stage (loaders, chain, arg-max waves), 0: the four-wave form of round 3
#define SYNC_ARGMAX10 1
#endif
#ifndef SYNC_UF96_MAX_WG
#define SYNC_UF96_MAX_WG 256
#endif
__device__ __forceinline__ void sy_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ void __launch_bounds__(256)
sync_metric_argmax4_kernel(float *__restrict__ cv, const float *__restrict__ corr, unsigned long long *__restrict__ keys, int n, int F, float alpha, int end_vec)
{
    __shared__ uint32_t ring[2][SYM_UF][64];
    // which of the four waves runs the chain rotates with the workgroup, so that the chain waves of the two or three workgroups a CU holds do
    // not sit on one SIMD (waves go to the SIMDs in order; workgroup b + 256 lands on the CU of workgroup b)
    const int lane = threadIdx.x & 63, chain = ((blockIdx.x >> 3) + (blockIdx.x >> 8)) & 3, wv = ((threadIdx.x >> 6) - chain) & 3;
    const int i = blockIdx.x * 64 + lane;
    const bool act = i < n;
    const int il = act ? i : n - 1;
    const int nq = (F + SYM_UF - 1) / SYM_UF;
    if (wv == 0) {
        float c = act ? cv[il] : 0.f;
        // positions past the last full vector are not averaged (:284-285): 0 c + 1 m = m exactly
        // (lanes past the last position carry 0)
        const float al = il < end_vec ? alpha : 0.f, om = !act ? 0.f : il < end_vec ? 1.0f - alpha : 1.f;
        float ma[SYM_UF], mb[SYM_UF];
        const char *pl = reinterpret_cast<const char *>(corr);        // wave-uniform pointer of the next frame to load, the lane's offset in 32 bits
        const unsigned ob = 4u * (unsigned)il;
        const size_t stride = sizeof(float) * (size_t)n;
        // The loads of the steady state are unconditional (a chunk that does not exist re-reads chunk 0), so that the loads of a chunk leave as
        // one batch and the wait before a chunk's arithmetic leaves the next chunk's loads in flight
        auto load = [&](float (&m)[SYM_UF], bool there) {
            const char *p = there ? pl : reinterpret_cast<const char *>(corr);
#pragma unroll
            for (int k = 0; k < SYM_UF; k++) { m[k] = *reinterpret_cast<const float *>(p + ob); p += stride; }
            if (there) pl = p;
        };
        auto run = [&](const float (&m)[SYM_UF], int q) {
#pragma unroll
            for (int k = 0; k < SYM_UF; k++) { c = al * c + om * m[k]; ring[q & 1][k][lane] = __float_as_uint(c); }
            sy_lds_barrier();
        };
        const int nfull = F / SYM_UF, rem = F - nfull * SYM_UF;
        if (nfull > 0) load(ma, true);
        int q = 0;
        for (; q + 2 <= nfull; q += 2) {
            load(mb, true);                                            // chunk q + 1
            run(ma, q);
            load(ma, q + 2 < nfull);                                   // chunk q + 2 if it is a whole one
            run(mb, q + 1);
        }
        if (q < nfull) run(ma, q++);                                   // an odd number of whole chunks: the last one is in ma
        if (rem > 0) {                                                 // the frames of a partial last chunk, one at a time
            for (int k = 0; k < rem; k++) {
                const float m = __builtin_nontemporal_load(reinterpret_cast<const float *>(pl + ob));
                pl += stride;
                c = al * c + om * m;
                ring[q & 1][k][lane] = __float_as_uint(c);
            }
            sy_lds_barrier();
        }
        if (act) cv[i] = c;
    } else {
        for (int q = 0; q < nq; q++) {
            sy_lds_barrier();                                          // chunk q is in ring[q & 1]
            const int cnt = F - q * SYM_UF < SYM_UF ? F - q * SYM_UF : SYM_UF;
            uint32_t b[8], w[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { b[j] = ring[q & 1][3 * j + wv - 1][lane]; w[j] = b[j]; }      // (slots past cnt hold older frames: not used below)
            sy_wave_umax4(w[0], w[1], w[2], w[3]);
            sy_wave_umax4(w[4], w[5], w[6], w[7]);
            int hi = 0, lo = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t wk = (uint32_t)__builtin_amdgcn_readlane((int)w[j], 63);
                const unsigned long long eq = __ballot(b[j] == wk);
                const uint32_t idx = (uint32_t)(blockIdx.x * 64 + __ffsll((long long)eq) - 1);
                asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"(wk), "n"(j));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"(0xffffffffu - idx), "n"(j));
            }
            const int k = 3 * lane + wv - 1;                           // lane j < 8 holds the key of frame slot 3 j + wv - 1
            if (lane < 8 && k < cnt) keys[(size_t)(q * SYM_UF + k) * gridDim.x + blockIdx.x] = hi != 0 ? ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo : 0ull;
        }
    }
}

// ---- (round 4) the same stage with the chain wave's instruction stream cut to what only it can do, and many more frames between two barriers.
// The four-wave form above spends ~56 cycles per frame on the chain wave -- not on its two dependent operations (alpha c, then the add; 4.3 cycles each) but on everything
// else that wave issues per frame at a lone wave's one instruction per ~4.5 cycles (the load and its address, (1 - alpha) m, the LDS store), and on what a barrier costs per
// 24 frames (LDS round trips and the barrier itself: ~500 cycles, measured with one role at a time): 96 us for the 4096 frames of a 32APSK-S call, more than the correlators
// and the delay line together.  Here a workgroup is 1 + UF / 8 waves for 64 positions and a tick (the stretch between two barriers) is UF frames:
//   * UF / 16 LOADER waves, sixteen frames of every chunk each, their loads in flight for two ticks (two register sets), scale by (1 - alpha) and leave the chunk in
//     LDS as [lane][frame];
//   * ONE CHAIN wave reads four frames per ds_read_b128 (half a chunk ahead of the arithmetic), does the two dependent operations per frame in place and writes four
//     averages per ds_write_b128 out of registers that nothing overwrites before the tick is over -- 2.5 instructions per frame; its tick is its own LDS traffic
//     (~1280 cycles for the 24 + 24 pieces of a chunk) plus its arithmetic (~825): tools/probe_lds128.hip, profiles/r04_probe_lds128.txt;
//   * UF / 16 ARG-MAX waves take the previous chunk: lane (frame, part) scans 16 of the 64 positions of its frame out of LDS (strictly greater: the first maximum, as
//     the reference's scan), the four parts of a frame merge their 64-bit keys inside their quad (two DPP exchanges) -- ~7 instructions per frame instead of the ~29 of
//     the wave-wide DPP ladders.
// Same operations in the same order per position: al * c, om * m, their sum, each rounded once.  Used when the call has few positions (one workgroup per CU at
// most: 100 KB of LDS); long frames, where the stage is bound by its 4 bytes per sample anyway, keep the four-wave form.
#ifndef SYM_NC             // chain waves per workgroup (1, 2 or 4: 64 / NC positions each; measured 47.8 / 49.8 / 52.1 us -- an LDS piece does not get cheaper with fewer active lanes)
#define SYM_NC 1
#endif
#ifndef SYM_ABL            // timing-only ablations (wrong results): 1 no loads, 2 no arg max, 4 no chain arithmetic
#define SYM_ABL 0
#endif
#ifdef SYM_PROF
#define SYB() do { const unsigned long long t_a = __builtin_amdgcn_s_memtime(); busy += t_a - t_last; sy_lds_barrier(); t_last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SYB() sy_lds_barrier()
#endif
// (round 6) THE FRAMES OF A CALL IN SEGMENTS.  The average over frames is a first-order recurrence c_f = al c_(f-1) + om m_f per position: a serial chain over the call's
// frames that one wave per 64 positions walks while most of the chip idles (32APSK-S: 54 workgroups).  A recurrence of this form composes: over a segment of frames,
// c_end = A c_start + B with A = al^len and B = the segment's own result from c_start = 0.  So: a first kernel (sync_seg_partial_kernel) forms {A, B} of every segment but the
// last, all segments side by side; then every segment runs THIS kernel on its own workgroups (grid y), its start value folded from the carried average and the {A, B} of the
// segments before it.  Same recurrence, same operations inside a segment; across a seam the start value is rounded differently from the one-wave walk (al^len c + B instead of
// len steps): ~1e-7 relative, far inside the metric's 2e-4 bar.  DVBS2HIP_SYNC_SEGMENTS=1 keeps the single walk.
// {A, B} per SUB-SEGMENT of SYNC_SUB frames (a segment = seg_F / SYNC_SUB of them): one lane per position walks 64 frames with all of their loads in flight -- thousands of
// short walks side by side instead of a few long ones (the first form, one walk of 1024 frames per segment with 8 loads in flight, cost 49 us more than it saved)
__global__ void __launch_bounds__(64)
sync_seg_partial_kernel(const float *__restrict__ corr, float *__restrict__ seg, int n, int F, float alpha, int end_vec)
{
    const int i = blockIdx.x * 64 + (int)threadIdx.x, k = blockIdx.y;
    if (i >= n) return;
    const int f0 = k * SYNC_SUB, cnt = f0 + SYNC_SUB <= F ? SYNC_SUB : F - f0;
    const float al = i < end_vec ? alpha : 0.f, om = i < end_vec ? 1.0f - alpha : 1.f;
    const float *p = corr + (size_t)f0 * n + i;
    float a = 1.f, b = 0.f;
    if (cnt == SYNC_SUB) {
#pragma unroll
        for (int h = 0; h < SYNC_SUB; h += 32) {
            float m[32];
#pragma unroll
            for (int u = 0; u < 32; u++) m[u] = __builtin_nontemporal_load(p + (size_t)(h + u) * n);
#pragma unroll
            for (int u = 0; u < 32; u++) { const float x = om * m[u]; b = al * b + x; a = a * al; }
        }
    } else
        for (int f = 0; f < cnt; f++) { const float x = om * __builtin_nontemporal_load(p + (size_t)f * n); b = al * b + x; a = a * al; }
    seg[(size_t)(2 * k) * n + i] = a;
    seg[(size_t)(2 * k + 1) * n + i] = b;
}

template <int UF>
__global__ void __launch_bounds__(64 * (SYM_NC + UF / 8))
sync_metric_argmax_kernel(float *__restrict__ cv, const float *__restrict__ corr, unsigned long long *__restrict__ keys, int n, int F, float alpha, int end_vec,
                          const float *__restrict__ seg, int seg_F)
{
    // this workgroup's segment of the call's frames (grid y; one segment = the whole call when seg is null)
    const int seg_id = (int)blockIdx.y, n_seg = (int)gridDim.y, f_base = seg_id * seg_F;
    corr += (size_t)f_base * n;
    keys += (size_t)f_base * gridDim.x;
    F = F - f_base < seg_F ? F - f_base : seg_F;
    constexpr int ST = UF + 4;           // LDS row of a lane: UF frames + 4 words (16-byte pieces of the 64 lanes on all banks: ST / 4 is odd)
    constexpr int NL = UF / 16, NC = SYM_NC;
    static_assert(UF == 96 && ((ST / 4) & 1) == 1, "role placement below is written for 13 waves");
    __shared__ __attribute__((aligned(16))) float pm[2][64][ST];          // (1 - alpha) m of a chunk, [lane][frame]
    // the averaged metric of a chunk, [lane][frame] with 16 words of padding behind every 16 rows: the arg-max waves' four parts of a frame (rows 16 part + j) then
    // fall on different banks (without it all four hit one bank: their conflicted reads kept the LDS queue busy while the chain wave waited for its own)
    constexpr int RG = 16 * ST + 16;
    __shared__ __attribute__((aligned(16))) uint32_t ring[2][4 * RG];
    // (roles in wave order: chain, loaders, arg-max waves.  Placing the roles by SIMD -- the chain wave with three loaders, the arg-max waves on the other three SIMDs -- measured
    // 57 against 52.5 us; three more waves that leave at once, so that the chain wave has its SIMD to itself: 55.1 against 54.8; the other roles asleep for the first 256 /
    // 512 / 768 cycles of a tick, so that the chain wave's LDS reads go first: 47.2 / 47.8 / 49.9 against 45.9)
    const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;
    const bool act = i < n;
    const int il = act ? i : n - 1;
    const int nq = (F + UF - 1) / UF;
#ifdef SYM_PROF
    unsigned long long busy = 0ull, t_last = __builtin_amdgcn_s_memtime(), t_rd = 0ull, t_ar = 0ull;
    const unsigned long long t_begin = t_last;
#endif
    typedef float sy_f4 __attribute__((ext_vector_type(4)));
    typedef uint32_t sy_u4 __attribute__((ext_vector_type(4)));
    if (wv < NC) {
        // ---- chain (NC = 1: one wave; the split over NC waves of 64 / NC positions each is kept as a knob, it measured slower)
        __builtin_amdgcn_s_setprio(3);
        const int pc = (64 / NC) * wv + (lane & (64 / NC - 1)), ic = blockIdx.x * 64 + pc;
        const bool mine = lane < 64 / NC, actc = ic < n;
        const int ilc = actc ? ic : n - 1;
        float c = mine && actc ? cv[ilc] : 0.f;
        for (int j = 0; j < seg_id * (seg_F / SYNC_SUB); j++) c = mine && actc ? seg[(size_t)(2 * j) * n + ilc] * c + seg[(size_t)(2 * j + 1) * n + ilc] : 0.f;      // the sub-segments before this segment, folded
        const float al = ilc < end_vec ? alpha : 0.f;         // positions past the last full vector are not averaged (:284-285): 0 c + 1 m = m exactly
        SYB();                                                // chunk 0 is in pm[0]
        for (int t = 0; t <= nq; t++) {
            if (t < nq && !(SYM_ABL & 4) && mine) {
                const sy_f4 *src = reinterpret_cast<const sy_f4 *>(&pm[t & 1][pc][0]);
                sy_u4 *dst = reinterpret_cast<sy_u4 *>(&ring[t & 1][(pc >> 4) * RG + (pc & 15) * ST]);
                const int cnt = F - t * UF;
                if (cnt >= UF) {
                    // The chain wave's tick is its LDS traffic, not its arithmetic (tools/probe_lds128.hip: a lone wave's ds_read_b128 of 64 x 16 bytes takes ~20 cycles,
                    // a ds_write_b128 ~34; 24 + 24 of them = 1280 cycles against 825 for the 192 dependent operations), so the two run side by side: half of the chunk's
                    // reads first, then group after group { arithmetic in place, its write, the read twelve groups ahead }, in this order exactly (sched_barrier: left to
                    // the scheduler a read sinks to just in front of its use and a write to just in front of the instruction that reuses its registers -- and a
                    // ds_write_b128 whose data registers are overwritten holds the vector unit, so every group keeps registers of its own until the tick is over).
                    constexpr int G = UF / 4, AH = G / 2;
                    sy_f4 in[G];
                    sy_u4 r[G];
#pragma unroll
                    for (int g = 0; g < AH; g++) in[g] = src[g];
                    __builtin_amdgcn_sched_barrier(0);
#ifdef SYM_PROF
                    const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
                    t_rd += tq0 - t_last;
#endif
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        c = al * c + in[g].x; r[g].x = __float_as_uint(c);
                        c = al * c + in[g].y; r[g].y = __float_as_uint(c);
                        c = al * c + in[g].z; r[g].z = __float_as_uint(c);
                        c = al * c + in[g].w; r[g].w = __float_as_uint(c);
                        __builtin_amdgcn_sched_barrier(0);
                        dst[g] = r[g];
                        if (g + AH < G) in[g + AH] = src[g + AH];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int g = 0; g < G; g++) asm volatile("" :: "v"(r[g]));
#ifdef SYM_PROF
                    t_ar += __builtin_amdgcn_s_memtime() - tq0;
#endif
                } else {                                      // the partial last chunk: the chain stops at the call's last frame
                    for (int g = 0; 4 * g < cnt; g++) {
                        const sy_f4 cur = src[g];
                        sy_u4 r;
                        c = al * c + cur.x; r.x = __float_as_uint(c);
                        if (4 * g + 1 < cnt) c = al * c + cur.y;
                        r.y = __float_as_uint(c);
                        if (4 * g + 2 < cnt) c = al * c + cur.z;
                        r.z = __float_as_uint(c);
                        if (4 * g + 3 < cnt) c = al * c + cur.w;
                        r.w = __float_as_uint(c);
                        dst[g] = r;
                    }
                }
            }
            SYB();
        }
        if (mine && actc && seg_id == n_seg - 1) cv[ic] = c;
    } else if (wv < NC + NL) {
        // ---- loaders: frames 16 L .. 16 L + 15 of every chunk; the loads of chunk t + 3 leave in tick t and are taken up in tick t + 2
        const int L = wv - NC;
        const float om = !act ? 0.f : il < end_vec ? 1.0f - alpha : 1.f;      // (lanes past the last position carry 0)
        const size_t stride = (size_t)n;
        float ma[16], mb[16];
        auto issue = [&](float (&m)[16], int chunk) {
            const int f0 = chunk * UF + 16 * L;
            const float *p = corr + (size_t)f0 * stride + il;
            if (SYM_ABL & 1) {
#pragma unroll
                for (int k = 0; k < 16; k++) m[k] = (float)(f0 + k);
            } else if (f0 + 16 <= F) {
#pragma unroll
                for (int k = 0; k < 16; k++) m[k] = __builtin_nontemporal_load(p + (size_t)k * stride);
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) m[k] = f0 + k < F ? __builtin_nontemporal_load(p + (size_t)k * stride) : 0.f;      // (wave-uniform test)
            }
        };
        auto stage = [&](const float (&m)[16], int chunk) {
            sy_f4 *dst = reinterpret_cast<sy_f4 *>(&pm[chunk & 1][lane][16 * L]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                sy_f4 a;
                a.x = om * m[4 * g]; a.y = om * m[4 * g + 1]; a.z = om * m[4 * g + 2]; a.w = om * m[4 * g + 3];
                dst[g] = a;
            }
        };
        // chunk c travels in set c & 1
        issue(ma, 0);
        if (nq > 1) issue(mb, 1);
        stage(ma, 0);
        if (nq > 2) issue(ma, 2);
        SYB();
        for (int t = 0; t <= nq; t += 2) {
            if (t + 1 < nq) { stage(mb, t + 1); if (t + 3 < nq) issue(mb, t + 3); }
            SYB();
            if (t + 1 > nq) break;
            if (t + 2 < nq) { stage(ma, t + 2); if (t + 4 < nq) issue(ma, t + 4); }
            SYB();
        }
    } else {
        // ---- arg max of the chunk the chain wave finished one barrier ago: sixteen frames per wave, lane = (frame, part of the 64 positions)
        const int R = wv - NC - NL, part = lane & 3;
        const int fr = 16 * R + (lane >> 2);
        const unsigned nwg = gridDim.x;
        SYB();
        for (int t = 0; t <= nq; t++) {
            if (t >= 1 && !(SYM_ABL & 2)) {
                const int q = t - 1;
                const int cnt = F - q * UF < UF ? F - q * UF : UF;
                if (16 * R < cnt) {
                    const uint32_t *src = &ring[q & 1][part * RG + fr];
                    uint32_t best = 0u, bi = 0u, v[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) v[j] = src[j * ST];
                    __builtin_amdgcn_sched_barrier(0);      // all sixteen reads in flight before the first compare (left alone, the scheduler waits for them pair by pair)
#pragma unroll
                    for (int j = 0; j < 16; j++) { const bool g = v[j] > best; best = g ? v[j] : best; bi = g ? (uint32_t)j : bi; }
                    uint32_t hi = best, lo = best != 0u ? 0xffffffffu - (uint32_t)(blockIdx.x * 64 + 16 * part + (int)bi) : 0u;
#pragma unroll
                    for (int st = 0; st < 2; st++) {
                        const uint32_t ohi = st == 0 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, false) : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xf, 0xf, false);
                        const uint32_t olo = st == 0 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xf, 0xf, false) : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xf, 0xf, false);
                        const bool tk = ohi > hi || (ohi == hi && olo > lo);
                        hi = tk ? ohi : hi; lo = tk ? olo : lo;
                    }
                    if (part == 0 && fr < cnt) keys[(size_t)(q * UF + fr) * nwg + blockIdx.x] = ((unsigned long long)hi << 32) | lo;
                }
            }
            SYB();
        }
    }
#ifdef SYM_PROF
    if (blockIdx.x == 1 && lane == 0) printf("wv %d wave %d busy %llu total %llu ticks %d rd %llu ar %llu\n", wv, (int)(threadIdx.x >> 6), busy, (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin), nq, t_rd, t_ar);
#endif
}
#undef SYB

// per frame (one wave each): the maximum of the workgroups' keys -> delay (:296), TRI = get_metric() (Synchronizer_frame.hxx:181),
// FLG = get_packet_flag() (.hpp:60); the delay line's table D[f] = 2 ((n - delay[f]) % n) (set_delay((cplx_in_sz - delay) % cplx_in_sz), :298);
// max_corr of the last frame into the handle's state
__global__ void __launch_bounds__(64)
sync_finalize_kernel(const unsigned long long *__restrict__ keys, int nwg, int32_t *__restrict__ delay, float *__restrict__ metric, int32_t *__restrict__ flag,
                     float trigger, int32_t *__restrict__ Dtab, float *__restrict__ last_metric, int n, int n_sof, int n_plsc, int F)
{
    const int f = blockIdx.x;
    unsigned long long k = 0ull;
    for (int g = threadIdx.x; g < nwg; g += 64) { const unsigned long long v = keys[(size_t)f * nwg + g]; k = v > k ? v : k; }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { const unsigned long long o = __shfl_xor(k, sft); k = o > k ? o : k; }
    if (threadIdx.x != 0) return;
    const uint32_t hi = (uint32_t)(k >> 32);
    const float bv = __uint_as_float(hi);
    const int idx = hi ? (int)(0xffffffffu - (uint32_t)k) : 0;       // no value above 0: index 0, as the reference's initial max_idx
    const int d = (n + idx - n_sof - n_plsc) % n;
    delay[f] = d;
    metric[f] = bv;
    if (flag) flag[f] = bv > trigger ? 1 : 0;
    if (f == F - 1) *last_metric = bv;
    Dtab[f] = 2 * ((n - d) % n);
}

// ---- Variable_delay_cc_naive::_filter (Variable_delay_cc_naive.cpp:56-79), the whole batch in ONE launch.
// The reference runs frame after frame: output f is built by four copies (a later one overwrites an earlier one; what
// none of them covers keeps the previous content of the output buffer) from input f, the delay line buff2 and output
// f - 1, and buff2[0 .. D_f) := the tail of input f.  Every one of these is a COPY, so each output sample is one
// particular earlier input sample (or a sample of the state the call started with, or the zero of first_time), and
// which one follows from the delays alone:
//   * buff2 after frame g holds, at index k, the tail of the LAST frame g' <= g whose D_g' exceeds k (vd_buff);
//   * an output sample that the reference takes from output f - 1 is resolved by looking at frame f - 1 in turn (vd_source).
// In lock (constant delay) neither walk takes a step; while the delay moves they take one or two.  One lane per output
// sample, no frame-to-frame launches: the per-frame form cost 4 us per frame (a launch each) whatever the frame size.
// st = {head2, first_time}; Dmax = the largest D of the call, which bounds vd_buff's walk.
// D[f] = 2 * ((n - delay[f]) % n) (set_delay((cplx_in_sz - delay) % cplx_in_sz), :298), tabulated per frame by sync_finalize_kernel:
// the walks below would otherwise pay two integer divisions per output sample
__device__ __forceinline__ int vd_D(const int32_t *Dtab, int f, int) { return Dtab[f]; }

// (every D, head2, nbuff2 and frame size is an even number of floats, so a complex sample never straddles two of the copies: the
// walks run once per complex sample and move 8 bytes).  The walks only compute WHERE a sample comes from (they read nothing but the
// small table of delays); the loads themselves are issued afterwards, several per lane at once -- one dependent load per lane was
// latency-bound at 1.6 TB/s.  A null source stands for the zero of first_time.
__device__ __noinline__ const float *vd_buff(const float *__restrict__ X, const float *__restrict__ buff0, const int32_t *__restrict__ delay_f,
                                                int g, int k, int n)
{
    const int N = 2 * n;
    for (; g >= 0; g--) {
        const int Dg = vd_D(delay_f, g, n);
        if (k < Dg) return &X[(size_t)g * N + N - Dg + k];
    }
    return &buff0[k];
}

__device__ __noinline__ const float *vd_source(const float *__restrict__ X, const float *__restrict__ yprev0, const float *__restrict__ buff0,
                                           const int *__restrict__ st0, const int32_t *__restrict__ delay_f, int f, int j, int n, int nbuff2)
{
    const int N = 2 * n;
    for (;;) {
        const int D = vd_D(delay_f, f, n), head2 = f == 0 ? st0[0] : vd_D(delay_f, f - 1, n), first = f == 0 ? st0[1] : 0;
        if (j >= D) return &X[(size_t)f * N + j - D];
        const int start_Y = D > head2 ? D - head2 : 0, start_buff = D < head2 ? head2 - D : 0;
        int end_buff = start_buff + D;
        end_buff = end_buff > nbuff2 ? nbuff2 : end_buff;
        end_buff = (end_buff - start_buff > N - start_Y) ? end_buff - ((end_buff - start_buff) - (N - start_Y)) : end_buff;
        if (j >= start_Y && j < start_Y + (end_buff - start_buff)) return vd_buff(X, buff0, delay_f, f - 1, start_buff + j - start_Y, n);
        if (j < start_Y) { if (first) return nullptr; j = N - start_Y + j; }
        if (--f < 0) return &yprev0[j];                     // the output buffer as the previous call left it
    }
}

// The first step of vd_source / vd_buff on values that are uniform over the workgroup (D of this frame, D of the one before = head2, the
// bounds of the four copies): no load before the sample's own.  In lock this resolves every sample (a whole output frame is one run of the
// input stream, X[f N - D ...)); what it cannot resolve goes through the general walks above.
struct VdU { int D, head2, first, start_Y, len, start_buff; };
__device__ __forceinline__ VdU vd_uniform(const int *__restrict__ st0, const int32_t *__restrict__ Dtab, int f, int n, int nbuff2)
{
    VdU u;
    const int N = 2 * n;
    u.D = Dtab[f]; u.head2 = f == 0 ? st0[0] : Dtab[f - 1]; u.first = f == 0 ? st0[1] : 0;
    u.start_Y = u.D > u.head2 ? u.D - u.head2 : 0; u.start_buff = u.D < u.head2 ? u.head2 - u.D : 0;
    int end_buff = u.start_buff + u.D;
    end_buff = end_buff > nbuff2 ? nbuff2 : end_buff;
    end_buff = (end_buff - u.start_buff > N - u.start_Y) ? end_buff - ((end_buff - u.start_buff) - (N - u.start_Y)) : end_buff;
    u.len = end_buff - u.start_buff;
    return u;
}
__device__ __forceinline__ const float *vd_source_fast(const VdU &u, const float *__restrict__ X, const float *__restrict__ yprev0, const float *__restrict__ buff0,
                                                       const int *__restrict__ st0, const int32_t *__restrict__ Dtab, int f, int j, int n, int nbuff2)
{
    const int N = 2 * n;
    if (j >= u.D) return &X[(size_t)f * N + j - u.D];
    if (f > 0 && j >= u.start_Y && j < u.start_Y + u.len) {
        const int k = u.start_buff + j - u.start_Y;
        if (k < u.head2) return &X[(size_t)(f - 1) * N + N - u.head2 + k];
    }
    return vd_source(X, yprev0, buff0, st0, Dtab, f, j, n, nbuff2);
}

constexpr int VD_SPL = 2;        // PAIRS of complex samples per lane, their loads in flight together
typedef float vd_f4 __attribute__((ext_vector_type(4), aligned(4)));     // a pair of complex samples wherever it lies (the delays are even numbers of floats only)
// blockIdx.y < F: output frame blockIdx.y; blockIdx.y == F: the delay line and {head2, first_time} after the last frame.
// A lane resolves two neighbouring complex samples; almost always they are neighbours at the source too (in lock a whole output frame is ONE
// run of the input stream, X[f N - D ...)), and the pair then moves as 16 bytes.
__global__ void sync_vdelay_batch_kernel(const float *__restrict__ X, const float *__restrict__ yprev0, float *__restrict__ yprev_new, float *__restrict__ Y,
                                         const float *__restrict__ buff_old, float *__restrict__ buff_new, const int *__restrict__ st_old,
                                         int *__restrict__ st_new, const int32_t *__restrict__ delay_f, int n, int nbuff2, int F, const int32_t *__restrict__ list)
{
    const int N = 2 * n;
    __shared__ int red[256];
    // located form (sync_locate_kernel): `list` = {count, frames to materialize .. (F = the state row among them)}; row y of the grid takes entries y, y + gridDim.y, ..
    const int n_list = list ? list[0] : 1;
    for (int li = list ? (int)blockIdx.y : 0; li < n_list; li += list ? (int)gridDim.y : 1) {
    const int f = list ? list[1 + li] : (int)blockIdx.y;
    int dmax = 0;
    if (f == F) {                                                     // the largest D of the call (a few thousand ints out of L2)
        for (int g = threadIdx.x; g < F; g += 256) dmax = delay_f[g] > dmax ? delay_f[g] : dmax;
        red[threadIdx.x] = dmax;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s]; __syncthreads(); }
        dmax = red[0];
    }
    const int lim = f < F ? N : nbuff2;
    const VdU u = vd_uniform(st_old, delay_f, f < F ? f : F - 1, n, nbuff2);
    {
    const int bx = blockIdx.x;
    const float *sa[VD_SPL], *sb[VD_SPL];
    int jj[VD_SPL];
#pragma unroll
    for (int i = 0; i < VD_SPL; i++) {
        const int j = 4 * ((bx * VD_SPL + i) * (int)blockDim.x + (int)threadIdx.x);      // first float of a pair of complex samples
        jj[i] = j; sa[i] = nullptr; sb[i] = nullptr;
        if (f < F) {
            if (j < lim) sa[i] = vd_source_fast(u, X, yprev0, buff_old, st_old, delay_f, f, j, n, nbuff2);
            if (j + 2 < lim) sb[i] = vd_source_fast(u, X, yprev0, buff_old, st_old, delay_f, f, j + 2, n, nbuff2);
        } else {                                                      // buff2 after the last frame: its first D entries are that frame's tail
            if (j < lim) sa[i] = j < u.D ? &X[(size_t)(F - 1) * N + N - u.D + j] : j < dmax ? vd_buff(X, buff_old, delay_f, F - 2, j, n) : &buff_old[j];
            if (j + 2 < lim) sb[i] = j + 2 < u.D ? &X[(size_t)(F - 1) * N + N - u.D + j + 2] : j + 2 < dmax ? vd_buff(X, buff_old, delay_f, F - 2, j + 2, n) : &buff_old[j + 2];
        }
    }
    vd_f4 v[VD_SPL];
#pragma unroll
    for (int i = 0; i < VD_SPL; i++) {
        if (sa[i] && sb[i] == sa[i] + 2) v[i] = *reinterpret_cast<const vd_f4 *>(sa[i]);
        else {
            const float2 lo = sa[i] ? *reinterpret_cast<const float2 *>(sa[i]) : make_float2(0.f, 0.f);       // a null source stands for the zero of first_time
            const float2 hi = sb[i] ? *reinterpret_cast<const float2 *>(sb[i]) : make_float2(0.f, 0.f);
            v[i] = vd_f4{lo.x, lo.y, hi.x, hi.y};
        }
    }
    float *dst = f < F ? Y + (size_t)f * N : buff_new;
#pragma unroll
    for (int i = 0; i < VD_SPL; i++) {
        const int j = jj[i];
        if (j + 2 < lim) {
            *reinterpret_cast<vd_f4 *>(dst + j) = v[i];
            if (f == F - 1) *reinterpret_cast<vd_f4 *>(yprev_new + j) = v[i];                    // the output buffer as the NEXT call finds it
        } else if (j < lim) {
            *reinterpret_cast<float2 *>(dst + j) = make_float2(v[i].x, v[i].y);
            if (f == F - 1) *reinterpret_cast<float2 *>(yprev_new + j) = make_float2(v[i].x, v[i].y);
        }
    }
    }
    if (f == F && blockIdx.x == 0 && threadIdx.x == 0) { st_new[0] = vd_D(delay_f, F - 1, n); st_new[1] = 0; }
    __syncthreads();      // (`red` is reused by the next entry of the list)
    }
}

// ================================================================ fine frequency / phase synchronizers
// Synchronizer_Luise_Reggiannini_DVBS2_aib (src/common/Module/Synchronizer/Synchronizer_freq/Synchronizer_freq_fine/
// Synchronizer_Luise_Reggiannini_DVBS2_aib.cpp:93-167) and Synchronizer_freq_phase_DVBS2_aib (.cpp:44-112): an estimate
// from the 36-symbol pilot blocks of a PL-descrambled frame (one small workgroup per frame, the sums in the reference's
// order), then a rotation of the whole frame.  Pilot p starts at symbol 1530 + 1476 p (.cpp:18-24).
constexpr int SFF_MAXP = 64;

// ---- L&R, per frame: tR[f] = (temp_R_l_0, temp_R_l_1) (.cpp:102-130)
__global__ void __launch_bounds__(256)
sff_lr_pilot_kernel(const float *__restrict__ X, float2 *__restrict__ tR, float2 *__restrict__ fr_not_yet, int n)
{
    __shared__ float2 part[SFF_MAXP * 9];
    const float *x = X + (size_t)blockIdx.x * 2 * n;
    int P = 0;
    for (int idx = 1530; idx < n; idx += 1476) P++;
    const int Lp = 18;
    for (int w = threadIdx.x; w < P * 9; w += 256) {
        const int p = w / 9, m = w % 9 + 1, ps = 1530 + 1476 * p;
        float s0 = 0.f, s1 = 0.f;
        for (int k = m; k < Lp; k++) {
            const float ar = x[2 * (ps + k)], ai = x[2 * (ps + k) + 1], br = x[2 * (ps + k - m)], bi = x[2 * (ps + k - m) + 1];
            const float zkr = ar + ai, zki = ai - ar, zmr = br + bi, zmi = bi - br;       // z = x (1 - j), :107-114
            s0 += zkr * zmr + zki * zmi;
            s1 += zki * zmr - zkr * zmi;
        }
        part[w] = make_float2(s0 / (float)(2 * (Lp - m)), s1 / (float)(2 * (Lp - m)));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f;
        for (int w = 0; w < P * 9; w++) { t0 += part[w].x; t1 += part[w].y; }             // p-major, m-minor: the reference's order
        tR[blockIdx.x] = make_float2(t0, t1);
        if (fr_not_yet) fr_not_yet[blockIdx.x] = make_float2(__uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu));      // sff_lr_fused_kernel's SFF_NOT_YET
    }
}

// ---- L&R, the damped autocorrelation from frame to frame (:131-135); fr[f] = {estimated_freq, est * pi}.  One wave:
// 64 frames at a time are staged in LDS (the next 64 already on their way from memory), lane 0 runs the recurrence over them (two
// multiply-adds per frame, the only serial part: 16 frames' values are read from LDS ahead of their dependent operations, so the chain
// pays one LDS round trip per 16 frames instead of one per frame), then every lane forms the estimate of one frame (atan2 and the
// double-precision scaling, in parallel).
template <bool FULL>
__device__ __forceinline__ void sff_lr_chain(float2 *sh, float &r0, float &r1, float alpha, float one_m, int cnt)
{
#pragma unroll 1
    for (int k0 = 0; k0 < 64; k0 += 16) {
        if (!FULL && k0 >= cnt) break;
        float2 t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = sh[k0 + k];
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (FULL || k0 + k < cnt) {
                r0 = alpha * r0 + one_m * t[k].x;
                r1 = alpha * r1 + one_m * t[k].y;
                t[k] = make_float2(r0, r1);
            }
#pragma unroll
        for (int k = 0; k < 16; k++) sh[k0 + k] = t[k];           // (entries past cnt are not read again)
    }
}

// Two waves: wave 0 stages 64 frames in LDS (the next 64 already on their way from memory) and its lane 0 runs the recurrence over them;
// wave 1 forms the estimates of the batch before (atan2 and the double-precision scaling, one frame per lane) meanwhile -- one LDS-only
// barrier per batch, two batch buffers.
__global__ void __launch_bounds__(128)
sff_lr_iir_kernel(const float2 *__restrict__ tR, float *__restrict__ R_l, float2 *__restrict__ fr, float *__restrict__ FRQ,
                  float *__restrict__ PHS, int F, float alpha)
{
    __shared__ float2 sh[2][64];
    const int lane = threadIdx.x & 63, nb = (F + 63) / 64;
    if (threadIdx.x < 64) {
        float r0 = R_l[0], r1 = R_l[1];                           // (lane 0's copy is the one that counts)
        const float one_m = 1 - alpha;
        float2 nxt = lane < F ? tR[lane] : make_float2(0.f, 0.f);
        for (int b = 0; b < nb; b++) {
            const int f = b * 64 + lane, cnt = F - b * 64 < 64 ? F - b * 64 : 64;
            sh[b & 1][lane] = nxt;
            if (f + 64 < F) nxt = tR[f + 64];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) { if (cnt == 64) sff_lr_chain<true>(sh[b & 1], r0, r1, alpha, one_m, cnt); else sff_lr_chain<false>(sh[b & 1], r0, r1, alpha, one_m, cnt); }
            sy_lds_barrier();                                     // batch b is in sh[b & 1]
        }
        if (lane == 0) { R_l[0] = r0; R_l[1] = r1; }
    } else {
        for (int b = 0; b < nb; b++) {
            sy_lds_barrier();
            const int f = b * 64 + lane;
            if (f < F) {
                const float2 r = sh[b & 1][lane];
                float est = atan2f(r.y, r.x);
                est = (float)((double)est / ((18 / 2 + 1) * 3.1415926535897932384626433832795));
                fr[f] = make_float2(est, (float)((double)est * 3.1415926535897932384626433832795));
                if (FRQ) FRQ[f] = est;
                if (PHS) PHS[f] = 0.f;
            }
        }
    }
}

// ---- freq_phase, per frame: fr[f] = {estimated_freq, estimated_phase} (.cpp:53-101)
__global__ void __launch_bounds__(64)
sff_fp_pilot_kernel(const float *__restrict__ X, float2 *__restrict__ fr, float *__restrict__ FRQ, float *__restrict__ PHS, int n)
{
    __shared__ float phase_est[SFF_MAXP];
    const float *x = X + (size_t)blockIdx.x * 2 * n;
    int P = 0;
    for (int idx = 1530; idx < n; idx += 1476) P++;
    const int Lp = 36;
    const double PI = 3.1415926535897932384626433832795;
    for (int p = threadIdx.x; p < P; p += 64) {
        const int ps = 1530 + 1476 * p;
        float s0 = 0.f, s1 = 0.f;
        for (int i = 0; i < Lp; i++) { s0 += x[2 * ps + 2 * i] + x[2 * ps + 2 * i + 1]; s1 += x[2 * ps + 2 * i + 1] - x[2 * ps + 2 * i]; }
        float ph = atan2f(s1, s0);
        phase_est[p] = ph < 0 ? (float)(ph + 2 * PI) : ph;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv_2PI = (float)(1.0f / (2 * PI));
        float acc = 0.f, sum_t = 0.f, sum_y = 0.f, sum_ty = 0.f, sum_tt = 0.f;
        for (int p = 0; p < P; p++) {
            float y;
            if (p == 0) y = inv_2PI * phase_est[0];
            else {
                const float diff_angle = phase_est[p] - phase_est[p - 1];
                float acc_elt = diff_angle > 0 ? floorf(diff_angle * inv_2PI + 0.5f) : ceilf(diff_angle * inv_2PI - 0.5f);
                acc_elt = (double)fabsf(diff_angle) > PI ? acc_elt : 0.0f;
                acc += acc_elt;
                y = inv_2PI * phase_est[p] - acc;
            }
            const float t = (float)(1530 + 1476 * p) + (float)(Lp / 2);
            sum_t += t; sum_y += y; sum_ty += t * y; sum_tt += t * t;
        }
        const float ef = (P * sum_ty - sum_t * sum_y) / (P * sum_tt - sum_t * sum_t);
        const float ep = (sum_y - ef * sum_t) / P;
        fr[blockIdx.x] = make_float2(ef, ep);
        if (FRQ) FRQ[blockIdx.x] = ef;
        if (PHS) PHS[blockIdx.x] = ep;
    }
}

// ---- rotation of the frame: y = x conj(e^{j theta}); MODE 0 (L&R, :141-166): theta = (est pi) * (2 k);
// MODE 1 (freq_phase, :103-112): theta = 2 pi (freq k + phase), formed in double as the reference's expression is
typedef float vd_f4n __attribute__((ext_vector_type(4)));      // a 16-byte type the non-temporal builtins accept
template <int MODE>
__device__ __forceinline__ float2 sff_rotate(float2 v, float2 e, int k)
{
    float theta;
    if (MODE == 0) theta = e.y * (float)(2 * k);
    else theta = (float)(2 * 3.1415926535897932384626433832795 * (double)(e.x * (float)k + e.y));
    float sn, c;
    sincosf(theta, &sn, &c);
    return make_float2(v.x * c + v.y * sn, v.y * c - v.x * sn);
}
template <int MODE>
__global__ void sff_rotate_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, const float2 *__restrict__ fr, int n, long long n_total)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_total) return;
    const int f = (int)(g / n), k = (int)(g - (long long)f * n);
    y[g] = sff_rotate<MODE>(x[g], fr[f], k);
}
// even frame lengths (every DVB-S2 PL frame) and 16-byte aligned sockets: two samples of one frame per lane, 16 bytes per access
template <int MODE>
__global__ void sff_rotate2_kernel(const vd_f4n *__restrict__ x, vd_f4n *__restrict__ y, const float2 *__restrict__ fr, int n, long long n_pairs)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_pairs) return;
    const int f = (int)((2 * g) / n), k = (int)(2 * g - (long long)f * n);
    const float2 e = fr[f];
    const vd_f4n v = __builtin_nontemporal_load(x + g);
    const float2 a = sff_rotate<MODE>(make_float2(v.x, v.y), e, k), b = sff_rotate<MODE>(make_float2(v.z, v.w), e, k + 1);
    __builtin_nontemporal_store(vd_f4n{a.x, a.y, b.x, b.y}, y + g);
}
template <int MODE>
static void sff_rotate_launch(const float *X, float *Y, const float2 *fr, int n, long long tot, hipStream_t s)
{
    if (n % 2 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0)
        hipLaunchKernelGGL(sff_rotate2_kernel<MODE>, dim3((unsigned)((tot / 2 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const vd_f4n *>(X),
                           reinterpret_cast<vd_f4n *>(Y), fr, n, tot / 2);
    else
        hipLaunchKernelGGL(sff_rotate_kernel<MODE>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2 *>(X),
                           reinterpret_cast<float2 *>(Y), fr, n, tot);
}

static hipError_t sff_lr_launch_unfused(const float *X, float *Y, float *R_l, float *tmp, float *FRQ, float *PHS, int n, int F, float alpha, hipStream_t s);
// the waiting workgroups' limit in ticks of s_memrealtime (100 MHz): one second; DVBS2HIP_LR_TIMEOUT_US overrides it (tests: 0 makes every waiting workgroup give up at
// its first unsuccessful poll, which exercises the error word and the recovery)
static unsigned long long sff_lr_timeout_ticks()
{
    const char *ev = getenv("DVBS2HIP_LR_TIMEOUT_US");
    return ev ? (unsigned long long)atoll(ev) * 100ull : 100000000ull;
}

// ---- L&R, recurrence and rotation in ONE launch.  The recurrence over the frames of a call is serial (4096 frames: 50 us in its own kernel, during which the
// machine idled, then a launch gap, then the rotation); here workgroup 0 runs it while every other workgroup rotates 1024 sample pairs of one frame: it requests
// its samples, then waits for its frame's estimate.  An estimate is published as ONE 8-byte word per frame (a relaxed agent-scope store; sff_lr_pilot_kernel has
// left SFF_NOT_YET in every word), so a reader needs no fence -- an agent-scope acquire per workgroup invalidates its XCD's L2 and was measured at ~100 ns per
// WORKGROUP of the launch, ten times the three kernels -- and no counter is shared by the workgroups.  Workgroup 0 is the first one the dispatcher places (an
// ASSUMPTION about the dispatcher, not a HIP guarantee: INTEGRATION.md "L&R dispatch order"); a workgroup that does not see its estimate within one second of
// wall-clock time (s_memrealtime, 100 MHz) sets the handle's error word (host-mapped: the host reads it at its next synchronisation point, no copy), drops its
// stores and ends -- the queue drains, the context survives, and since workgroup 0 has published every estimate by the time the launch is over the host only has
// to run the rotation again (sff_lr_recover).  The recurrence itself on two lanes (lane 0 the real, lane 1
// the imaginary part: two dependent instructions per frame instead of six), its (1 - alpha) t products formed by all lanes before; the values and their order are
// those of sff_lr_iir_kernel (alpha r + (1 - alpha) t, two roundings).
constexpr int SFF_RU = 4, SFF_RCH = 256 * SFF_RU;        // 16-byte pairs per lane / per workgroup
constexpr unsigned long long SFF_NOT_YET = ~0ull;        // (an estimate that is not a number is stored as the canonical NaN: never this pattern)
template <bool FULL>
__device__ __forceinline__ void sff_lr_chain1(float *pl, float &r, float alpha, int cnt)
{
    if (FULL) {
        // a whole batch: its 64 values are requested at once (one LDS round trip per batch in front of the 128 dependent instructions instead of one per 16 frames)
        float t[64];
#pragma unroll
        for (int k = 0; k < 64; k++) t[k] = pl[k];
#pragma unroll
        for (int k = 0; k < 64; k++) { r = alpha * r + t[k]; t[k] = r; }
#pragma unroll
        for (int k = 0; k < 64; k++) pl[k] = t[k];
        return;
    }
#pragma unroll 1
    for (int k0 = 0; k0 < 64; k0 += 16) {
        if (k0 >= cnt) break;
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = pl[k0 + k];
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k0 + k < cnt) { r = alpha * r + t[k]; t[k] = r; }
#pragma unroll
        for (int k = 0; k < 16; k++) pl[k0 + k] = t[k];
    }
}
__global__ void __launch_bounds__(256)
sff_lr_fused_kernel(const vd_f4n *__restrict__ x, vd_f4n *__restrict__ y, const float2 *__restrict__ tR, float *__restrict__ R_l, float2 *fr, float *__restrict__ FRQ,
                    float *__restrict__ PHS, int F, float alpha, int n, int cpf, uint32_t *err, unsigned long long timeout_ticks)
{
    __shared__ __attribute__((aligned(16))) float sh[2][2][64];        // [batch parity][component][frame of the batch]
    __shared__ float2 s_e;
    __shared__ int s_drop;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0) {
        if (threadIdx.x >= 128) return;                                 // (waves that have ended do not count at the barriers below)
        __builtin_amdgcn_s_setprio(3);                                  // the one serial thread of the launch: ahead of the rotating waves on its SIMD
        const int nb = (F + 63) / 64;
        if (threadIdx.x < 64) {
            float r = lane < 2 ? R_l[lane] : 0.f;
            const float one_m = 1 - alpha;
            float2 nxt = lane < F ? tR[lane] : make_float2(0.f, 0.f);
            for (int b = 0; b < nb; b++) {
                const int f = b * 64 + lane, cnt = F - b * 64 < 64 ? F - b * 64 : 64;
                sh[b & 1][0][lane] = one_m * nxt.x;
                sh[b & 1][1][lane] = one_m * nxt.y;
                if (f + 64 < F) nxt = tR[f + 64];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane < 2) { if (cnt == 64) sff_lr_chain1<true>(sh[b & 1][lane], r, alpha, cnt); else sff_lr_chain1<false>(sh[b & 1][lane], r, alpha, cnt); }
                sy_lds_barrier();                                     // batch b is in sh[b & 1]
            }
            if (lane < 2) R_l[lane] = r;
        } else {
            for (int b = 0; b < nb; b++) {
                sy_lds_barrier();
                const int f = b * 64 + lane;
                if (f < F) {
                    const float r0 = sh[b & 1][0][lane], r1 = sh[b & 1][1][lane];
                    float est = atan2f(r1, r0);
                    est = (float)((double)est / ((18 / 2 + 1) * 3.1415926535897932384626433832795));
                    const float estpi = (float)((double)est * 3.1415926535897932384626433832795);
                    const uint32_t w0 = est == est ? __float_as_uint(est) : 0x7FC00000u, w1 = estpi == estpi ? __float_as_uint(estpi) : 0x7FC00000u;
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(fr + f), (unsigned long long)w0 | ((unsigned long long)w1 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (FRQ) FRQ[f] = est;
                    if (PHS) PHS[f] = 0.f;
                }
            }
        }
        return;
    }
    const int w = (int)blockIdx.x - 1, f = w / cpf, c = w - f * cpf, npf = n / 2;
    const vd_f4n *xf = x + (size_t)f * npf;
    vd_f4n *yf = y + (size_t)f * npf;
    const int p0 = c * SFF_RCH + (int)threadIdx.x;
    vd_f4n v[SFF_RU];
#pragma unroll
    for (int u = 0; u < SFF_RU; u++) { const int q = p0 + u * 256; v[u] = __builtin_nontemporal_load(xf + (q < npf ? q : npf - 1)); }
    if (threadIdx.x == 0) {
        unsigned long long e;
        uint32_t nap = 1;
        int drop = 0;
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        while ((e = __hip_atomic_load(reinterpret_cast<unsigned long long *>(fr + f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == SFF_NOT_YET) {
            for (uint32_t i = 0; i < nap; i++) __builtin_amdgcn_s_sleep(1);
            if (nap < 64) nap *= 2;
            if (__builtin_amdgcn_s_memrealtime() - t_start > timeout_ticks) {       // the recurrence is not running (it cannot happen with in-order dispatch): report, do not trap
                if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                drop = 1;
                break;
            }
        }
        s_drop = drop;
        s_e = make_float2(__uint_as_float((uint32_t)e), __uint_as_float((uint32_t)(e >> 32)));
    }
    __syncthreads();
    if (s_drop) return;                                                 // nothing of this workgroup's stretch is stored: sff_lr_recover rotates the call again
    const float2 e = s_e;
#pragma unroll
    for (int u = 0; u < SFF_RU; u++) {
        const int q = p0 + u * 256;
        if (q < npf) {
            const float2 a = sff_rotate<0>(make_float2(v[u].x, v[u].y), e, 2 * q), b = sff_rotate<0>(make_float2(v[u].z, v[u].w), e, 2 * q + 1);
            __builtin_nontemporal_store(vd_f4n{a.x, a.y, b.x, b.y}, yf + q);
        }
    }
}

// after a launch whose error word was set: every estimate is in `tmp` (workgroup 0 has ended), only rotated samples are missing -- rotate the whole call again
hipError_t sff_lr_recover(const float *X, float *Y, float *tmp, int n, int F, hipStream_t s)
{
    float2 *fr = reinterpret_cast<float2 *>(tmp) + F;
    sff_rotate_launch<0>(X, Y, fr, n, (long long)n * F, s);
    return hipGetLastError();
}

hipError_t sff_lr_launch(const float *X, float *Y, float *R_l, float *tmp /* 4 F floats */, float *FRQ, float *PHS, int n, int F, float alpha, uint32_t *err_dev, hipStream_t s)
{
    const char *ev = getenv("DVBS2HIP_LR");                    // read at every call
    const bool unfused = ev && !strcmp(ev, "unfused");
    if (!unfused && n % 2 == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0) {
        float2 *tR = reinterpret_cast<float2 *>(tmp), *fr = tR + F;
        const int cpf = (n / 2 + SFF_RCH - 1) / SFF_RCH;
        hipLaunchKernelGGL(sff_lr_pilot_kernel, dim3(F), dim3(256), 0, s, X, tR, fr, n);
        hipLaunchKernelGGL(sff_lr_fused_kernel, dim3(1 + (unsigned)F * cpf), dim3(256), 0, s, reinterpret_cast<const vd_f4n *>(X), reinterpret_cast<vd_f4n *>(Y), tR, R_l, fr, FRQ, PHS,
                           F, alpha, n, cpf, err_dev, sff_lr_timeout_ticks());
        return hipGetLastError();
    }
    return sff_lr_launch_unfused(X, Y, R_l, tmp, FRQ, PHS, n, F, alpha, s);
}

static hipError_t sff_lr_launch_unfused(const float *X, float *Y, float *R_l, float *tmp, float *FRQ, float *PHS, int n, int F, float alpha, hipStream_t s)
{
    float2 *tR = reinterpret_cast<float2 *>(tmp), *fr = tR + F;
    hipLaunchKernelGGL(sff_lr_pilot_kernel, dim3(F), dim3(256), 0, s, X, tR, (float2 *)nullptr, n);
    hipLaunchKernelGGL(sff_lr_iir_kernel, dim3(1), dim3(128), 0, s, tR, R_l, fr, FRQ, PHS, F, alpha);
    const long long tot = (long long)n * F;
    sff_rotate_launch<0>(X, Y, fr, n, tot, s);
    return hipGetLastError();
}

hipError_t sff_fp_launch(const float *X, float *Y, float *tmp /* 2 F floats */, float *FRQ, float *PHS, int n, int F, hipStream_t s)
{
    float2 *fr = reinterpret_cast<float2 *>(tmp);
    hipLaunchKernelGGL(sff_fp_pilot_kernel, dim3(F), dim3(64), 0, s, X, fr, FRQ, PHS, n);
    const long long tot = (long long)n * F;
    sff_rotate_launch<1>(X, Y, fr, n, tot, s);
    return hipGetLastError();
}

std::vector<uint16_t> sync_frag_default() { return sync_mfma_frag(K_CONJ_SOF, K_CONJ_PLSC); }

hipError_t sync_corr_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, float *cor_sof, float *cor_plsc, long long n_total, hipStream_t s)
{
    // frag: the correlators on the matrix cores (k_sync_mfma.hip, which also leaves the next call's memory in xh_out); null = the fp32 vector kernel
    if (sync_mfma_usable(x, frag)) return sync_corr_mfma_launch(x, xh_in, xh_out, frag, cor_sof, cor_plsc, n_total, s);
    const unsigned grid = (unsigned)((n_total + SY_T - 1) / SY_T);
    hipLaunchKernelGGL(sync_corr_kernel, dim3(grid), dim3(SY_THREADS), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(xh_in),
                       reinterpret_cast<float2 *>(cor_sof), reinterpret_cast<float2 *>(cor_plsc), n_total);
    hipLaunchKernelGGL(sync_hist_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(xh_in),
                       reinterpret_cast<float2 *>(xh_out), SY_H, n_total);
    return hipGetLastError();
}

// the average over frames with the arg max of every frame, then the per-frame sockets / delay-line table
static void sync_average_argmax(float *cv, float *corr, int n, int F, float alpha, int vec_width, const SyncTail &t, hipStream_t s)
{
    const int end_vec = (n / vec_width) * vec_width, nwg = (n + 63) / 64;
    // few positions (short frames): the thirteen-wave form, one workgroup per CU at most; long frames: round 3's four-wave form (the stage is bound by its 4 bytes per
    // sample there: QPSK-N 1024 frames 42-45 us in the four-wave form, 65 us in this one)
    static const bool four = getenv("DVBS2HIP_SYNC_ARGMAX4") != nullptr;      // development: round 3's kernel for every frame length
    if (SYNC_ARGMAX10 && !four && nwg <= SYNC_UF96_MAX_WG) {
        // segments of the call's frames side by side while they fit the chip (one workgroup per CU) and stay long enough to be worth a seam (>= 768 frames each)
        static const int cap = getenv("DVBS2HIP_SYNC_SEGMENTS") ? atoi(getenv("DVBS2HIP_SYNC_SEGMENTS")) : SYNC_MAX_SEG;      // (development: 1 = the single walk)
        int S = cap <= 1 || !t.seg ? 1 : std::min(std::min(std::min(cap, SYNC_MAX_SEG), SYNC_UF96_MAX_WG / nwg), F / 768);
        int seg_F = F;
        if (S > 1) { seg_F = ((F + S - 1) / S + 191) / 192 * 192; S = (F + seg_F - 1) / seg_F; }      // (a multiple of the chunk's 96 frames and of SYNC_SUB)
        if (S > 1) hipLaunchKernelGGL(sync_seg_partial_kernel, dim3(nwg, (S - 1) * (seg_F / SYNC_SUB)), dim3(64), 0, s, (const float *)corr, t.seg, n, F, alpha, end_vec);
        else { S = 1; seg_F = F; }
        hipLaunchKernelGGL(sync_metric_argmax_kernel<96>, dim3(nwg, S), dim3(64 * (SYM_NC + 12)), 0, s, cv, corr, t.keys, n, F, alpha, end_vec, (const float *)t.seg, seg_F);
    }
    else hipLaunchKernelGGL(sync_metric_argmax4_kernel, dim3(nwg), dim3(256), 0, s, cv, corr, t.keys, n, F, alpha, end_vec);
    hipLaunchKernelGGL(sync_finalize_kernel, dim3(F), dim3(64), 0, s, t.keys, nwg, t.delay, t.metric, t.flag, t.trigger, t.Dtab, t.last_metric, n, 25, 64, F);
}

// the one-task form: correlators + instantaneous metric fused (nothing but m leaves the chip), then the average / arg max as above
hipError_t sync_corr_metric_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, const float *sofh_in, float *sofh_out, float *cv, float *corr,
                                   const SyncTail &t, int n, int F, float alpha, int vec_width, hipStream_t s)
{
    const long long tot = (long long)n * F;
    if (sync_mfma_usable(x, frag)) (void)sync_corr_m_mfma_launch(x, xh_in, xh_out, frag, sofh_in, sofh_out, corr, tot, s);
    else {
        hipLaunchKernelGGL(sync_corr_m_kernel, dim3((unsigned)((tot + SY_T - 1) / SY_T)), dim3(SY_THREADS), 0, s, reinterpret_cast<const float2 *>(x),
                           reinterpret_cast<const float2 *>(xh_in), reinterpret_cast<const float2 *>(sofh_in), reinterpret_cast<float2 *>(sofh_out), corr, tot);
        hipLaunchKernelGGL(sync_hist_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<const float2 *>(xh_in),
                           reinterpret_cast<float2 *>(xh_out), SY_H, tot);
    }
    sync_average_argmax(cv, corr, n, F, alpha, vec_width, t, s);
    return hipGetLastError();
}

hipError_t sync_metric_launch(const float *cor_sof, const float *sofh_in, float *sofh_out, const float *cor_plsc, float *cv, float *corr,
                              const SyncTail &t, int n, int F, float alpha, int vec_width, hipStream_t s)
{
    const long long tot = (long long)n * F;
    hipLaunchKernelGGL(sync_m_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float2 *>(cor_sof),
                       reinterpret_cast<const float2 *>(sofh_in), reinterpret_cast<const float2 *>(cor_plsc), corr, tot);
    hipLaunchKernelGGL(sync_hist_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const float2 *>(cor_sof), reinterpret_cast<const float2 *>(sofh_in),
                       reinterpret_cast<float2 *>(sofh_out), 64, tot);
    sync_average_argmax(cv, corr, n, F, alpha, vec_width, t, s);
    return hipGetLastError();
}

// (round 5) The LOCATED form of the frame synchronizer: Variable_delay_cc_naive's output frame f is, whenever the delay did not move between f - 1 and f, ONE run of the
// input stream -- X[f N - D_f, f N - D_f + N) (vd_source_fast: samples j >= D come from input f, samples j < D from the tail of input f - 1, which the delay line holds) --
// so a consumer of ours can read it THERE and the shifted copy (a third of the synchronizer's time, a pure copy at 5.9 TB/s) is not made.  need[f] = 1 marks the frames
// that are NOT such a run -- frame 0 (its head is state of the previous call), a frame whose delay differs from its predecessor's, a delay beyond the line's length --
// and the last frame (the next call's `yprev`): those are materialized by sync_vdelay_batch_kernel into `scratch` (frame f at scratch + f N), src[f] points there.
__global__ void sync_locate_kernel(const float *X, float *scratch, const int32_t *__restrict__ Dtab, int n, int nbuff2, int F, int32_t *__restrict__ list, const float **__restrict__ src)
{
    // one workgroup; list[0] = count, list[1 ..] = the frames to materialize in any order, then F (the delay line's state row).  The count is kept in LDS while the list is
    // built (no memset of list[0] in front of the launch: a fill kernel of its own in every call).  (Doing this loop in the last workgroup of sync_finalize_kernel to leave
    // instead -- one launch less -- was measured: the F arrivals on one counter word cost 25 ns each, 4096 frames 0.100 -> 0.203 ms: docs/negative_results.md)
    __shared__ int cnt;
    const size_t N = 2 * (size_t)n;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const int D = Dtab[f];
        const bool run = f > 0 && f < F - 1 && D == Dtab[f - 1] && D <= nbuff2;
        src[f] = run ? X + (size_t)f * N - (size_t)D : scratch + (size_t)f * N;
        if (!run) list[1 + atomicAdd(&cnt, 1)] = f;
    }
    __syncthreads();
    if (threadIdx.x == 0) { list[1 + cnt] = F; list[0] = cnt + 1; }
}

// all F frames of a call; Dtab (F ints) comes from sync_finalize_kernel; Yprev_new = the last output frame for the next call.  need / src != null: the located form
// (Y = scratch for the frames that are materialized)
hipError_t sync_vdelay_launch(const float *X, const float *Yprev, float *Yprev_new, float *Y, const float *buff_old, float *buff_new, const int *st_old, int *st_new,
                              const int32_t *Dtab, int n, int nbuff2, int F, hipStream_t s, int32_t *list, const float **src)
{
    const int tot = ((nbuff2 > 2 * n ? nbuff2 : 2 * n) + 3) / 4;     // pairs of complex samples
    const int gx = (tot + 256 * VD_SPL - 1) / (256 * VD_SPL);
    if (list) {
        hipLaunchKernelGGL(sync_locate_kernel, dim3(1), dim3(1024), 0, s, X, Y, Dtab, n, nbuff2, F, list, src);
        // rows of the grid share the list's entries: in lock three of them have work (first frame, last frame, state row), while the synchronizer acquires all do
        const int gy = F + 1 < 64 ? F + 1 : 64;
        hipLaunchKernelGGL(sync_vdelay_batch_kernel, dim3(gx, gy), dim3(256), 0, s, X, Yprev, Yprev_new, Y, buff_old, buff_new, st_old, st_new, Dtab, n, nbuff2, F, (const int32_t *)list);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(sync_vdelay_batch_kernel, dim3(gx, F + 1), dim3(256), 0, s, X, Yprev, Yprev_new, Y, buff_old, buff_new, st_old, st_new, Dtab, n, nbuff2, F, (const int32_t *)nullptr);
    return hipGetLastError();
}

}  // namespace dvbs2
