// a5 -- SRRC matched filter on the matrix cores (BASELINE config 5: "FIR as batched GEMM"), gfx950.
//
// Same interface and semantics as fir_ccr_kernel (k_fir.hip), i.e. Filter_FIR_ccr<R>::_filter
// (/root/reference src/common/Module/Filter/Filter_FIR/Filter_FIR_ccr.cpp:68-142):
//     y[n] = sum_{k=0}^{T-1} brev[k] * x[n - (T-1) + k],   complex x, REAL taps, T <= 81.
//
// The stream is cut into blocks of 16 samples; block a of one plane (re or im -- real taps never mix them) is column a of
// X[j][a] = x[16 a + j], and the 16 outputs of block a are
//     Y[i][a] = sum_{k=0}^{95} A[i][k] * x[16 (a - 5) + k],      A[i][k] = brev81[k - i]  (0 <= k - i <= 80, else 0)
// -- a [16 x 96] banded Toeplitz matrix (84 % dense) times a [96 x 16] matrix whose columns are overlapping 96-sample
// windows, 16 blocks (256 consecutive outputs) per product: three v_mfma_f32_16x16x32_bf16 per plane and tile.
//
// fp32 MFMA runs at the vector-FMA rate on gfx950 (no gain), one bf16 product is 2^-9 accurate (far outside the 1e-4
// bar).  So both operands are split EXACTLY into three bf16 terms (x = x1 + x2 + x3, 8 + 8 + 8 significant bits, each
// remainder formed in fp32 without rounding) and the six products of weight >= 2^-24 are accumulated in the MFMA's fp32
// accumulator, smallest first:  b3 x1 + b2 x2 + b1 x3 + b2 x1 + b1 x2 + b1 x1.  Every bf16 x bf16 product is exact in
// fp32, so the only errors are the three dropped products (~2^-25 |b||x|) and the fp32 accumulation -- measured 1e-6 on
// unit-power input, the same as the reference's own fp32 FMA chain (whose rounding order this does not reproduce; parity
// bar 1e-4 as for the vector kernel, tests/test_fir_gpu.py).  18 MFMAs x 16 cycles per 256 real outputs = 0.56 of the
// cycles the v_pk_fma_f32 kernel needs at its peak, and the splitting is done once per sample while staging the tile
// into LDS (planar bf16: 12 B per complex sample), so the kernel is left with the HBM stream: 8 B in + 8 B out.
//
// Lane maps (cdna_hip_programming.md section 3): lane l = (c = l & 15, g = l >> 4) holds A[row c][k = 8 g + j] and
// B[k = 8 g + j][col c], j = 0..7, for each 32-wide K step; D[row 4 g + r][col c] in accumulator register r.  The
// product is formed transposed -- the sample windows are the A operand (row c = block c), the band the B operand (column c
// = output c of the block): a sample fragment is 8 consecutive bf16 of one plane (one ds_read_b128, 16-B aligned, conflict
// free as the lane groups of that instruction stand), and accumulator r of lane (c, g) is output c of block 4 g + r, so
// the 16 lanes of a row store one whole 128-B line.
#include "dvbs2hip_internal.h"
#include <vector>
#include <cstring>
#include <cstdlib>

namespace dvbs2 {

constexpr int FM_THREADS = 256;
constexpr int FM_TILE = 2048;                     // outputs per workgroup = 8 MFMA tiles of 256
constexpr int FM_H = 80;                          // history the Toeplitz band is laid out for (T = 81)
constexpr int FM_NS = FM_TILE + FM_H;             // staged samples per workgroup
// No padding: the 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md, LDS) read the 16-B slots
// 2 c + g + const of a plane, which are 16 different slots of the 256-B bank row for every group as the lanes stand
// (a pad of one slot per row measured 59 % conflict cycles); the 4-byte staging stores (one pair of samples per lane and plane) are contiguous.
__host__ __device__ constexpr int fm_pad(int i) { return i; }
constexpr int FM_PLANE = (fm_pad(FM_NS) + 7) & ~7;   // bf16 elements per plane

typedef __bf16 fm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fm_f32x4 __attribute__((ext_vector_type(4)));
typedef float fm_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t fm_pk(float a, float b)      // v_cvt_pk_bf16_f32, round to nearest even
{
    fm_f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fm_bf16x2));
}
// (a, b) -> three packed bf16 pairs with a = a1 + a2 + a3 exactly (likewise b)
__device__ __forceinline__ void fm_split(float a, float b, uint32_t &p1, uint32_t &p2, uint32_t &p3)
{
    p1 = fm_pk(a, b);
    a -= __uint_as_float(p1 << 16); b -= __uint_as_float(p1 & 0xffff0000u);
    p2 = fm_pk(a, b);
    a -= __uint_as_float(p2 << 16); b -= __uint_as_float(p2 & 0xffff0000u);
    p3 = fm_pk(a, b);
}

// LDS-only barrier: the global loads of the next tile stay in flight across it (a __syncthreads() would drain them)
__device__ __forceinline__ void fm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int FM_NPASS = FM_TILE / (2 * FM_THREADS);           // sample pairs per lane and tile
static_assert(FM_NPASS * 2 * FM_THREADS == FM_TILE && FM_H % 2 == 0 && 6 * (FM_H / 2) <= FM_THREADS, "tile = whole passes; the overlap = whole pairs");

// split one pair of consecutive samples into the three bf16 parts and store them at sample index o of the six planes
__device__ __forceinline__ void fm_stage(uint16_t *lds, int o, float re0, float im0, float re1, float im1)
{
    uint32_t p1, p2, p3;
    fm_split(re0, re1, p1, p2, p3);
    *reinterpret_cast<uint32_t *>(lds + 0 * FM_PLANE + o) = p1;
    *reinterpret_cast<uint32_t *>(lds + 2 * FM_PLANE + o) = p2;
    *reinterpret_cast<uint32_t *>(lds + 4 * FM_PLANE + o) = p3;
    fm_split(im0, im1, p1, p2, p3);
    *reinterpret_cast<uint32_t *>(lds + 1 * FM_PLANE + o) = p1;
    *reinterpret_cast<uint32_t *>(lds + 3 * FM_PLANE + o) = p2;
    *reinterpret_cast<uint32_t *>(lds + 5 * FM_PLANE + o) = p3;
}
// sample gi on its own: the filter memory before the stream, zero after it
__device__ __forceinline__ float2 fm_fetch_edge(const float2 *__restrict__ x, const float2 *__restrict__ hist_in, int H, long long n_total, long long gi)
{
    float2 v = make_float2(0.f, 0.f);
    if (gi < 0) { if (gi >= -(long long)H) v = hist_in[H + gi]; }
    else if (gi < n_total) v = x[gi];
    return v;
}

// Workgroups filter `tiles_per_wg` consecutive tiles of 2048 outputs each (chunks of 4 on long streams: see fir_mfma_launch).  The loads of tile i + 1 (one
// contiguous KB per wave instruction) are in flight -- in registers, no branch between them and their use -- while tile i
// goes through the matrix cores; the 80-sample overlap with the previous tile is taken from LDS, already split, so the
// stream is read from HBM exactly once.  The SAMPLES are the A operand (row = block of 16) and the Toeplitz band the B
// operand (column = output inside the block): accumulator register r of lane (c, g) is output 16 (4 g + r) + c, so the
// 16 lanes of a row write one whole 128-B line of interleaved (re, im) pairs.
// NPH = 1: the matched filter.  NPH = 2: the TX shaping filter (Filter_UPFIR_ccr_naive, osf = 2) -- two polyphase branches
// of at most 81 taps over the SAME staged samples, branch ph in the band fragments afrag[ph]; output 2 i + ph of input i,
// so a lane stores (re, im) of both branches as one 16-byte piece and 16 lanes a 256-byte run.  n_total counts INPUT samples.
template <int NPH>
__global__ void __launch_bounds__(FM_THREADS) __attribute__((amdgpu_waves_per_eu(NPH == 1 ? 4 : 3, NPH == 1 ? 4 : 3)))
fir_mfma_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, const float2 *__restrict__ hist_in, float2 *__restrict__ hist_out,
                const uint4 *__restrict__ afrag, int T, long long n_total, int tiles_per_wg)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[3 * 2 * FM_PLANE];      // [part][re | im][sample]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = T - 1;
    long long blk0 = (long long)blockIdx.x * tiles_per_wg * FM_TILE;
    // whole pairs inside the stream are loaded without a branch (a pair that is not takes the address of pair 0 and is
    // re-fetched sample by sample when it is staged: only in the stream's last tile)
    fm_f32x4 v[FM_NPASS];
    auto prefetch = [&](long long b0, long long lim) {
#pragma unroll
        for (int ps = 0; ps < FM_NPASS; ps++) {
            const long long gi = b0 + 2 * tid + ps * 2 * FM_THREADS;
            v[ps] = __builtin_nontemporal_load(reinterpret_cast<const fm_f32x4 *>(x + (gi + 2 <= lim ? gi : 0)));
        }
    };
    prefetch(blk0, n_total);
    // the filter memory of the next call = the last H samples of (old memory ++ x); nothing here writes x or hist_in
    if (blockIdx.x == 0 && tid < H) {
        const long long gi = n_total - H + tid;
        hist_out[tid] = gi >= 0 ? x[gi] : hist_in[H + gi];
    }
    // the 80 samples before the workgroup's first tile
    if (tid < FM_H / 2) {
        const long long gi = blk0 - FM_H + 2 * tid;
        const float2 s0 = fm_fetch_edge(x, hist_in, H, n_total, gi), s1 = fm_fetch_edge(x, hist_in, H, n_total, gi + 1);
        fm_stage(lds, 2 * tid, s0.x, s0.y, s1.x, s1.y);
    }
    // the Toeplitz fragments of the three tap parts (host-made, 9 KB, L2-resident).  A polyphase branch of the shaping filter has at most
    // 49 taps (upfir_mfma_usable), right-aligned in the 81-entry band: nothing of it falls into the first 32-wide K step, which NPH = 2
    // therefore leaves out -- a third of its products multiplied zeros, and the 24 registers of those fragments were what it spilled for
    constexpr int S0 = NPH == 2 ? 1 : 0, NS = 3 - S0;
    fm_bf16x8 A[NPH][3][NS];
#pragma unroll
    for (int ph = 0; ph < NPH; ph++)
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int s = 0; s < NS; s++) A[ph][p][s] = __builtin_bit_cast(fm_bf16x8, afrag[((ph * 3 + p) * 3 + s + S0) * 64 + lane]);
    const int c = lane & 15, g = lane >> 4;
    // lane's share of the overlap that is carried from tile to tile inside LDS: 6 planes x 40 pairs
    const int ov_dst = (tid / (FM_H / 2)) * FM_PLANE + 2 * (tid % (FM_H / 2)), ov_src = ov_dst + FM_TILE;
    const bool ov = tid < 6 * (FM_H / 2);

    for (int it = 0; it < tiles_per_wg && blk0 < n_total; it++, blk0 += FM_TILE) {
        // ---- split the staged samples into three bf16 parts, planar in LDS
#pragma unroll
        for (int ps = 0; ps < FM_NPASS; ps++) {
            const int gidx = 2 * tid + ps * 2 * FM_THREADS;
            float re0 = v[ps].x, im0 = v[ps].y, re1 = v[ps].z, im1 = v[ps].w;
            if (blk0 + gidx + 2 > n_total) {
                const float2 s0 = fm_fetch_edge(x, hist_in, H, n_total, blk0 + gidx), s1 = fm_fetch_edge(x, hist_in, H, n_total, blk0 + gidx + 1);
                re0 = s0.x; im0 = s0.y; re1 = s1.x; im1 = s1.y;
            }
            fm_stage(lds, FM_H + gidx, re0, im0, re1, im1);
        }
        fm_lds_barrier();
        prefetch(blk0 + FM_TILE, it + 1 < tiles_per_wg ? n_total : 0);      // unconditional (no copy of the old registers): past the end it reads pair 0

#pragma unroll
        for (int t2 = 0; t2 < FM_TILE / 256 / (FM_THREADS / 64); t2++) {
            const int tt = wv + t2 * (FM_THREADS / 64);             // tile of 256 outputs inside the workgroup's 2048
            const long long o0 = blk0 + 256 * tt;
            if (o0 >= n_total) break;
            fm_f32x4 acc[NPH][2];
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                fm_bf16x8 B[3][NS];
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int s = 0; s < NS; s++) {
                        const int idx = 16 * (c + 2 * (s + S0) + 16 * tt) + 8 * g;
                        B[p][s] = __builtin_bit_cast(fm_bf16x8, *reinterpret_cast<const uint4 *>(lds + (2 * p + pl) * FM_PLANE + idx));
                    }
                // (tap part, sample part), smallest products first
                constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int ph = 0; ph < NPH; ph++) {
                    fm_f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 6; q++)
#pragma unroll
                        for (int s = 0; s < NS; s++) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[TB[q]][s], A[ph][TA[q]][s], d, 0, 0, 0);
                    acc[ph][pl] = d;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const long long o = o0 + 16 * (4 * g + r) + c;
                if (o >= n_total) continue;
                if (NPH == 1) {
                    fm_f32x2 w; w.x = acc[0][0][r]; w.y = acc[0][1][r];
                    __builtin_nontemporal_store(w, reinterpret_cast<fm_f32x2 *>(y + o));
                } else {
                    fm_f32x4 w; w.x = acc[0][0][r]; w.y = acc[0][1][r]; w.z = acc[NPH - 1][0][r]; w.w = acc[NPH - 1][1][r];
                    __builtin_nontemporal_store(w, reinterpret_cast<fm_f32x4 *>(y + 2 * o));
                }
            }
        }
        uint32_t carry = 0u;
        if (ov) carry = *reinterpret_cast<const uint32_t *>(lds + ov_src);
        fm_lds_barrier();                                           // every wave is done reading the planes
        if (ov) *reinterpret_cast<uint32_t *>(lds + ov_dst) = carry;   // the next tile's first 80 samples
    }
}

static inline uint16_t bf16_rne(float v)
{
    uint32_t u; std::memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; std::memcpy(&v, &u, 4); return v; }

// A fragments [part][K step][lane][8] of the banded Toeplitz matrix A[i][k] = brev81[k - i], brev81 = the reversed taps
// right-aligned in 81 entries (a shorter filter is the same sum with leading zero taps)
std::vector<uint16_t> fir_mfma_afrag(const float *taps_rev, int T)
{
    std::vector<uint16_t> out((size_t)3 * 3 * 64 * 8, 0);
    if (T < 1 || T > FM_H + 1) return out;
    for (int s = 0; s < 3; s++)
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) {
                const int i = l & 15, k = 32 * s + 8 * (l >> 4) + j, kk = k - i - (FM_H + 1 - T);
                float v = (kk >= 0 && kk < T) ? taps_rev[kk] : 0.f;
                for (int p = 0; p < 3; p++) {
                    const uint16_t h = bf16_rne(v);
                    out[(((size_t)p * 3 + s) * 64 + l) * 8 + j] = h;
                    v -= bf16_f32(h);
                }
            }
    return out;
}

bool fir_mfma_usable(const float *x, const float *y, int T, long long n_total)
{
    return T >= 1 && T <= FM_H + 1 && n_total >= 2 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

// also leaves the next call's filter memory in hist_out (T - 1 samples)
hipError_t fir_mfma_launch(const float *x, float *y, const float *hist_in, float *hist_out, const uint16_t *afrag, int T, long long n_total, hipStream_t s)
{
    const long long n_tiles = (n_total + FM_TILE - 1) / FM_TILE;
    // Chunks of 4 tiles, handed to the CUs by the hardware dispatcher as workgroups retire: resident workgroups do not run at
    // the same speed (the arbiter favours the older waves), and one persistent workgroup per slot with an equal share of the
    // stream waited for the slowest -- same-box sweep at 68 M samples: 1024 workgroups (33 tiles each) 0.222 ms, 8192 (5 tiles)
    // 0.207 ms, 16384 0.211 ms.  Short streams still get one workgroup per tile.  DVBS2HIP_FIR_WGS caps the grid instead.
    static const int max_wg = [] { const char *e = getenv("DVBS2HIP_FIR_WGS"); return e ? atoi(e) : 0; }();
    const int tiles_per_wg = max_wg > 0 ? (int)((n_tiles + max_wg - 1) / max_wg) : (n_tiles >= 4096 ? 4 : (int)((n_tiles + 1023) / 1024));
    const unsigned grid = (unsigned)((n_tiles + tiles_per_wg - 1) / tiles_per_wg);
    hipLaunchKernelGGL(fir_mfma_kernel<1>, dim3(grid), dim3(FM_THREADS), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<float2 *>(y),
                       reinterpret_cast<const float2 *>(hist_in), reinterpret_cast<float2 *>(hist_out), reinterpret_cast<const uint4 *>(afrag), T, n_total,
                       tiles_per_wg);
    return hipGetLastError();
}

// ---- N2: the TX shaping filter (Filter_UPFIR_ccr_naive.cpp:52-66) with osf = 2 on the matrix cores: output i * 2 + ph =
// sum_m H[ph + 2 m] x[i - m], i.e. two ordinary FIRs of Hin + 1 = (T - 1) / 2 + 1 taps over the input.  Band fragments of
// both branches, [branch][part][K step][lane][8]:
std::vector<uint16_t> upfir_mfma_afrag(const float *taps, int T)
{
    const int Hin = (T - 1) / 2, Tb = Hin + 1;
    std::vector<uint16_t> out;
    if (T < 1 || Tb > FM_H + 1) return out;
    for (int ph = 0; ph < 2; ph++) {
        std::vector<float> brev(Tb, 0.f);                      // brev[k] = h_ph[Tb - 1 - k], h_ph[m] = H[ph + 2 m]
        for (int m = 0; m < Tb; m++) if (ph + 2 * m < T) brev[Tb - 1 - m] = taps[ph + 2 * m];
        const std::vector<uint16_t> a = fir_mfma_afrag(brev.data(), Tb);
        out.insert(out.end(), a.begin(), a.end());
    }
    return out;
}

bool upfir_mfma_usable(const float *x, const float *y, int T, int osf, long long n_in)
{
    // a branch of at most 49 taps: the kernel skips the band's first K step
    return osf == 2 && T >= 1 && (T - 1) / 2 + 1 <= FM_H + 1 - 32 && n_in >= 2 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

// hist = the last (T - 1) / 2 input samples
hipError_t upfir_mfma_launch(const float *x, float *y, const float *hist_in, float *hist_out, const uint16_t *afrag2, int T, long long n_in, hipStream_t s)
{
    const long long n_tiles = (n_in + FM_TILE - 1) / FM_TILE;
    const int tiles_per_wg = n_tiles >= 4096 ? 4 : (int)((n_tiles + 1023) / 1024);       // as fir_mfma_launch
    const unsigned grid = (unsigned)((n_tiles + tiles_per_wg - 1) / tiles_per_wg);
    hipLaunchKernelGGL(fir_mfma_kernel<2>, dim3(grid), dim3(FM_THREADS), 0, s, reinterpret_cast<const float2 *>(x), reinterpret_cast<float2 *>(y),
                       reinterpret_cast<const float2 *>(hist_in), reinterpret_cast<float2 *>(hist_out), reinterpret_cast<const uint4 *>(afrag2),
                       (T - 1) / 2 + 1, n_in, tiles_per_wg);
    return hipGetLastError();
}

}  // namespace dvbs2
