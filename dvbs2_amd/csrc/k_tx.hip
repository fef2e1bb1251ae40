// N1 -- TX mirror + AWGN channel on the device, so the Monte-Carlo BER loop of
// dvbs2_tx_rx_bb (/root/reference src/mains/TX_RX_BB/main.cpp:75-82) never leaves the GPU:
//   source.generate -> bb_scrambler.scramble -> BCH encode (Encoder_BCH_DVBS2.cpp:28-43)
//   -> LDPC encode (enc type "LDPC_DVBS2", DVBS2.cpp:427) -> interleave (DVBS2.cpp:451-476)
//   -> modulate (Modem_generic) -> Framer::generate (Framer.hxx:232-293)
//   -> Scrambler_PL::scramble (Scrambler_PL.hxx:61-78) -> Channel_AWGN (DVBS2.cpp:593-613)
// Test-signal generation, not the RX hot path; parity: bit-exact against the oracle TX for
// given payloads and sigma = 0 (tests/test_tx_gpu.py).  Noise: Philox4x32-10 + Box-Muller,
// counter = (symbol, frame, stream), so results do not depend on the launch geometry.
#include "dvbs2hip_internal.h"

namespace dvbs2 {

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key)
{
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * ctr.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * ctr.z;
        ctr = make_uint4((uint32_t)(p1 >> 32) ^ ctr.y ^ key.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ ctr.w ^ key.y, (uint32_t)p0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}

// ---------------------------------------------------------------- source + BB scramble (coalesced, one workgroup per frame)
__global__ void __launch_bounds__(256)
tx_src_kernel(const TxKParams p)
{
    extern __shared__ uint32_t wsm[];                          // payload, packed
    const int f = blockIdx.x, tid = threadIdx.x, K = p.K_bch;
    const int nwk = (K + 31) / 32, nw_out = (p.K_ldpc + 31) / 32;
    for (int w = tid; w < nwk; w += 256) {
        uint32_t word = 0u;
        if (p.info_in) {
            const int32_t *src = p.info_in + (size_t)f * K + 32 * w;
            for (int b = 0; b < 32 && 32 * w + b < K; b++) word |= ((uint32_t)src[b] & 1u) << b;
        } else {
            const uint4 r = philox4x32(make_uint4((uint32_t)(w >> 2), (uint32_t)f, 0u, 0u), make_uint2(p.seed_lo, p.seed_hi));
            word = (w & 3) == 0 ? r.x : (w & 3) == 1 ? r.y : (w & 3) == 2 ? r.z : r.w;
            if (32 * w + 32 > K) word &= (1u << (K - 32 * w)) - 1u;
        }
        wsm[w] = word;
        p.bch_cw[(size_t)f * nw_out + w] = word ^ p.prbs[w];    // Scrambler_BB (prbs tail bits are zero)
    }
    __syncthreads();
    if (p.info_out)
        for (int k = tid; k < K; k += 256) p.info_out[(size_t)f * K + k] = (int32_t)((wsm[k >> 5] >> (k & 31)) & 1u);
}

// ---------------------------------------------------------------- BCH parity (Encoder_BCH_DVBS2.cpp:28-43)
// The systematic encoder is a polynomial division: parity = (u(x) x^r) mod g(x), serial along the frame.  The division is
// linear, so the frame is cut into TX_BCH_SEG consecutive segments, one LANE each (a frame per 16 lanes): the lane divides
// its segment a byte per step through a 256-entry table of (v(x) x^r) mod g(x) held in LDS (r = N - K <= 192 parity bits
// in three 64-bit words), moves its remainder to the segment's place -- times x^(8 * bytes behind the segment) mod g, a
// linear map applied bit by bit from a host-made table of x^(b + 8 after_s) mod g -- and the 16 remainders are XORed.
// (One lane per whole frame, the first version, left 4096 lanes with 7184 dependent steps each: 0.90 ms of the 2.36 ms TX.)
__global__ void __launch_bounds__(64)
tx_bchpar_kernel(const TxKParams p)
{
    __shared__ unsigned long long T[256][3];
    __shared__ uint8_t brev[256];
    for (int i = threadIdx.x; i < 256; i += 64) {
        T[i][0] = p.bch_tab[3 * i]; T[i][1] = p.bch_tab[3 * i + 1]; T[i][2] = p.bch_tab[3 * i + 2];
        uint32_t r = 0; for (int b = 0; b < 8; b++) if (i >> b & 1) r |= 1u << (7 - b);
        brev[i] = (uint8_t)r;
    }
    __syncthreads();
    const int seg = threadIdx.x & (TX_BCH_SEG - 1);
    const int f = blockIdx.x * (64 / TX_BCH_SEG) + (threadIdx.x / TX_BCH_SEG);
    const bool live = f < p.n_frames;
    const int K = p.K_bch, r = p.K_ldpc - p.K_bch;
    const int nw_out = (p.K_ldpc + 31) / 32;
    uint32_t *cw = p.bch_cw + (size_t)(live ? f : 0) * nw_out;
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    const int tw = (r - 8) >> 6, ts = (r - 8) & 63;            // where the top byte of the remainder sits
    const unsigned long long m1 = r >= 128 ? ~0ull : r > 64 ? (1ull << (r - 64)) - 1ull : 0ull;
    const unsigned long long m2 = r >= 192 ? ~0ull : r > 128 ? (1ull << (r - 128)) - 1ull : 0ull;
    const int nbytes = K / 8, L = (nbytes + TX_BCH_SEG - 1) / TX_BCH_SEG;
    const int b0 = min(seg * L, nbytes), b1 = min(b0 + L, nbytes);
    if (live && b1 > b0) {
        // the message words of the segment, 8 at a time and one batch ahead: the division is a dependent chain (table look-up
        // -> XOR -> next look-up) and must not also wait for a global load every four bytes
        constexpr int WB = 8;
        const int w0 = b0 >> 2, w1 = (b1 - 1) >> 2;           // first / last word touched
        uint32_t nxt[WB];
#pragma unroll
        for (int k = 0; k < WB; k++) nxt[k] = cw[min(w0 + k, w1)];
        for (int wb = w0; wb <= w1; wb += WB) {
            uint32_t cur[WB];
#pragma unroll
            for (int k = 0; k < WB; k++) cur[k] = nxt[k];
#pragma unroll
            for (int k = 0; k < WB; k++) nxt[k] = cw[min(wb + WB + k, w1)];
#pragma unroll
            for (int k = 0; k < WB; k++) {
#pragma unroll
                for (int bb = 0; bb < 4; bb++) {
                    const int by = 4 * (wb + k) + bb;
                    if (by < b0 || by >= b1) continue;
                    const uint32_t raw = (cur[k] >> (bb * 8)) & 0xFFu;
                    const uint32_t top = (uint32_t)((tw == 0 ? s0 : tw == 1 ? s1 : s2) >> ts) & 0xFFu;
                    const uint32_t idx = top ^ brev[raw];
                    s2 = ((s2 << 8) | (s1 >> 56)) & m2; s1 = ((s1 << 8) | (s0 >> 56)) & m1; s0 <<= 8;
                    if (r <= 64) s0 &= (r == 64 ? ~0ull : (1ull << r) - 1ull);
                    s0 ^= T[idx][0]; s1 ^= T[idx][1]; s2 ^= T[idx][2];
                }
            }
        }
    }
    // to the segment's place: sum over the set bits b of the remainder of x^(b + 8 (nbytes - b1)) mod g
    if (b1 < nbytes) {
        const unsigned long long *P = p.bch_shift + (size_t)seg * r * 3;
        unsigned long long a0 = 0, a1 = 0, a2 = 0;
        for (int b0 = 0; b0 < r; b0 += 8) {               // r is a multiple of 8 (m t, m = 14 or 16); 24 table loads in flight
            unsigned long long q0[8], q1[8], q2[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { q0[k] = P[3 * (b0 + k)]; q1[k] = P[3 * (b0 + k) + 1]; q2[k] = P[3 * (b0 + k) + 2]; }
            const unsigned long long sw = b0 < 64 ? s0 >> b0 : b0 < 128 ? s1 >> (b0 - 64) : s2 >> (b0 - 128);      // 8 bits of the remainder (b0 is a multiple of 8)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const unsigned long long m = 0ull - ((sw >> k) & 1ull);
                a0 ^= q0[k] & m; a1 ^= q1[k] & m; a2 ^= q2[k] & m;
            }
        }
        s0 = a0; s1 = a1; s2 = a2;
    }
    for (int o = TX_BCH_SEG / 2; o > 0; o >>= 1) { s0 ^= __shfl_xor(s0, o); s1 ^= __shfl_xor(s1, o); s2 ^= __shfl_xor(s2, o); }
    if (!live || seg != 0) return;
    // parity, coefficient of x^(r-1) first (DVB-S2 order), appended at bit K of the packed frame
    uint32_t outw = (K & 31) ? cw[K >> 5] : 0u;
    for (int j = 0; j < r; j++) {
        const int d = r - 1 - j, k = K + j;
        const uint32_t b = d < 64 ? (uint32_t)(s0 >> d) & 1u : d < 128 ? (uint32_t)(s1 >> (d - 64)) & 1u : (uint32_t)(s2 >> (d - 128)) & 1u;
        outw |= b << (k & 31);
        if ((k & 31) == 31) { cw[k >> 5] = outw; outw = 0; }
    }
    if ((p.K_ldpc & 31) != 0) cw[nw_out - 1] = outw;
}

// ---------------------------------------------------------------- LDPC IRA encoder (ETSI EN 302 307 5.3.2)
// One workgroup per frame.  parity accumulator address (a + m q) mod M <=> check (r, t): the same circulant structure the decoder
// uses, i.e. accumulator row r (its 360 bits t) = XOR over the row's edges (bit-group g, shift t0) of bit-group g ROTATED by t0.
// Rows are 12 packed words: a lane forms one word of one row, an edge costs it two funnel shifts out of the packed info bits (the
// wrap of the 360-bit circle splits a window in two) instead of 32 single-bit gathers -- 1 / 16 of the instructions of one lane per check.
// Then p_c ^= p_{c-1} over c = q t + r: a running XOR of the rows (prefix over r inside a column) and an exclusive prefix over t of
// the column totals, both on packed words.
constexpr int ENC_W = (LDPC_Z + 31) / 32;                            // 12 words per 360-bit row, the last one holds 8 bits
__device__ __forceinline__ uint32_t enc_window(const uint32_t *info, int pos)      // 32 bits of the packed info from bit `pos` on
{
    return __funnelshift_r(info[pos >> 5], info[(pos >> 5) + 1], pos & 31);
}
__global__ void __launch_bounds__(LDPC_THREADS)
tx_ldpc_kernel(const TxKParams p)
{
    extern __shared__ uint32_t sm[];
    const int K = p.K_ldpc, M = p.N_ldpc - p.K_ldpc, q = M / LDPC_Z;
    const int nw_in = (K + 31) / 32, nw_out = (p.N_ldpc + 31) / 32;
    uint32_t *info = sm;                                     // nw_in words + one of padding (a window may start in the last word)
    uint32_t *prow = sm + nw_in + 1;                         // [r][ENC_W] packed parity rows
    uint32_t *excl = prow + q * ENC_W;                       // ENC_W words: exclusive prefix over t of the column totals
    uint32_t *tab = excl + ENC_W;                            // the layer table (t0 | group << 9 per entry): no global round trip per entry
    const int t = threadIdx.x, f = blockIdx.x;
    for (int w = t; w < q * p.enc_stride; w += LDPC_THREADS) tab[w] = p.enc_tab[w];
    const uint32_t *src = p.bch_cw + (size_t)f * nw_in;
    for (int w = t; w <= nw_in; w += LDPC_THREADS) info[w] = w < nw_in ? src[w] : 0u;
    __syncthreads();
    for (int task = t; task < q * ENC_W; task += LDPC_THREADS) {
        const int r = task / ENC_W, l = task - r * ENC_W;
        const int deg = p.enc_deg[r];
        const uint32_t *T = tab + r * p.enc_stride;
        uint32_t acc = 0u;
#pragma unroll 4
        for (int j = 0; j < deg; j++) {
            const uint32_t e = T[j];                                  // t0 | group << 9
            int m0 = 32 * l - (int)(e & 0x1FFu); m0 += m0 < 0 ? LDPC_Z : 0;      // source index of the word's first bit: (32 l - t0) mod 360
            const int base = (int)(e >> 9) * LDPC_Z, n1 = LDPC_Z - m0;              // n1 bits are left before the circle wraps
            uint32_t w = enc_window(info, base + m0);
            if (n1 < 32) w = (w & ((1u << n1) - 1u)) | (enc_window(info, base) << n1);
            acc ^= w;
        }
        prow[task] = l == ENC_W - 1 ? acc & ((1u << (LDPC_Z - 32 * (ENC_W - 1))) - 1u) : acc;
    }
    __syncthreads();
    if (t < ENC_W) {
        // prefix over r inside every column (12 lanes, one word of every row each), then the exclusive prefix over t of the column totals
        uint32_t x = 0u;
        for (int r = 0; r < q; r++) { x ^= prow[r * ENC_W + t]; prow[r * ENC_W + t] = x; }
        uint32_t incl = x;
        incl ^= incl << 1; incl ^= incl << 2; incl ^= incl << 4; incl ^= incl << 8; incl ^= incl << 16;      // inclusive prefix XOR inside the word
        uint32_t par = incl >> 31;                                    // parity of the whole word (full words only matter: the last one has no successor)
        uint32_t carry = 0u;
        for (int l = 0; l < ENC_W; l++) { const uint32_t pl = (uint32_t)__shfl((int)par, l); if (l < t) carry ^= pl; }
        excl[t] = (incl << 1) ^ (carry ? 0xFFFFFFFFu : 0u);
    }
    __syncthreads();
    uint32_t *dst = p.ldpc_cw + (size_t)f * nw_out;
    for (int w = t; w < nw_out; w += LDPC_THREADS) {
        uint32_t word = 0;
        const int i0 = 32 * w;
        if (i0 + 32 <= K) word = info[w];                     // systematic part, word-aligned
        else {
            // parity bit c = i - K sits at (r = c mod q, tt = c / q): stepped, one division per word
            int c = i0 >= K ? i0 - K : 0, tt = c / q, r = c - tt * q;
            for (int b = 0; b < 32; b++) {
                const int i = i0 + b;
                if (i >= p.N_ldpc) break;
                uint32_t bit;
                if (i < K) bit = (info[i >> 5] >> (i & 31)) & 1u;
                else { bit = ((prow[r * ENC_W + (tt >> 5)] ^ excl[tt >> 5]) >> (tt & 31)) & 1u; if (++r == q) { r = 0; tt++; } }
                word |= bit << b;
            }
        }
        dst[w] = word;
    }
}

// ---------------------------------------------------------------- interleave + modulate + frame + PL scramble + AWGN
// One lane per PAIR of PL symbols: one Philox4x32-10 block (four words) is exactly the two Box-Muller pairs the two symbols
// need, and the lane leaves with one 16-byte store.
// BPS > 0: bits per symbol known at compile time (the bit gathers of a symbol are issued together, not one dependent load
// after the other) and, with ITL, the column/row interleaver with as many columns as bits per symbol (every DVB-S2 one):
// interleaved bit k bps + b sits at natural position col(b) * n_rows + k -- no division.  BPS == 0: any configuration.
template <int BPS, bool ITL>
__device__ __forceinline__ float2 tx_symbol(const TxKParams &p, const float *cs, const uint32_t *cw, int i, int n_pil)
{
    // Branch-free for the compile-time BPS forms: header, pilot and data symbols take the same instructions (indices clamped, the
    // right value selected at the end) -- a wave nearly always holds one kind only, but every branch costs the scalar unit a
    // save / restore of the execution mask, and there were seventy of those per pair of symbols.
    // inverse of the RX map: position i-90 inside [16 slots data | 36 pilots] blocks
    const int j = i >= 90 ? i - 90 : 0, blk = j / (1440 + 36), off = j - blk * (1440 + 36);
    const bool pilot = blk < n_pil && off >= 1440;
    int k = blk < n_pil ? blk * 1440 + off : n_pil * 1440 + (j - n_pil * (1440 + 36));
    float2 y;
    if (BPS > 0) {
        k = pilot ? 0 : k;                                           // (any valid symbol: its value is not used)
        const int n_rows = p.N_ldpc / BPS;
        uint32_t wd[BPS > 0 ? BPS : 1];
#pragma unroll
        for (int b = 0; b < BPS; b++) {
            const int nat = ITL ? (p.itl_order == 0 ? b : BPS - 1 - b) * n_rows + k : k * BPS + b;
            wd[b] = cw[nat >> 5] >> (nat & 31);
        }
        int idx = 0;
#pragma unroll
        for (int b = 0; b < BPS; b++) idx |= (int)(wd[b] & 1u) << b;
        y = make_float2(cs[2 * idx], cs[2 * idx + 1]);
        if (pilot) y = make_float2(0.70710678118654752440f, 0.70710678118654752440f);     // pilot (Framer.hxx:252-260)
    } else if (pilot) y = make_float2(0.70710678118654752440f, 0.70710678118654752440f);
    else {
        int idx = 0;
        for (int b = 0; b < p.bps; b++) {
            // interleaved bit k*bps+b comes from natural position col*n_rows + row (column/row interleaver)
            int nat = k * p.bps + b;
            if (p.itl_cols > 1) { const int row = nat / p.itl_cols, c = nat - row * p.itl_cols; nat = (p.itl_order == 0 ? c : p.itl_cols - 1 - c) * (p.N_ldpc / p.itl_cols) + row; }
            idx |= (int)((cw[nat >> 5] >> (nat & 31)) & 1u) << b;
        }
        y = make_float2(cs[2 * idx], cs[2 * idx + 1]);
    }
    // multiply by exp(j pi/2 R) (Scrambler_PL.hxx:66-76, scr_flag = true): R = 1: (-y, x), 2: (-x, -y), 3: (y, -x) -- a swap and sign flips
    const int R = p.pl_seq[j] & 3;
    const float a = (R & 1) ? -y.y : y.x, b = (R & 1) ? y.x : y.y;
    y = (R & 2) ? make_float2(-a, -b) : make_float2(a, b);
    const int ih = i < 90 ? i : 0;
    const float2 h = make_float2(p.plh[2 * ih], p.plh[2 * ih + 1]);   // the PL header is neither scrambled nor modulated here
    return i < 90 ? h : y;
}

constexpr int TX_MOD_U = 4;                               // pairs of PL symbols per lane (independent: their loads overlap)
template <int BPS, bool ITL>
__global__ void __launch_bounds__(256)
tx_mod_kernel(const TxKParams p)
{
    __shared__ float cs[64];
    const int f = blockIdx.y;
    if (threadIdx.x < (2 << p.bps)) cs[threadIdx.x] = p.cstl[threadIdx.x];
    __syncthreads();
    const int n_pil = p.n_sym / 1440;
    const uint32_t *cw = p.ldpc_cw + (size_t)f * ((p.N_ldpc + 31) / 32);
    const float sg = p.sigma ? p.sigma[f] : 0.f;
#pragma unroll
    for (int u = 0; u < TX_MOD_U; u++) {
        const int pr = (blockIdx.x * TX_MOD_U + u) * blockDim.x + threadIdx.x;     // pair of PL symbols 2 pr, 2 pr + 1
        const int i0 = 2 * pr;
        if (i0 >= p.pl_frame) continue;
        const bool two = i0 + 1 < p.pl_frame;
        float2 y0 = tx_symbol<BPS, ITL>(p, cs, cw, i0, n_pil), y1 = two ? tx_symbol<BPS, ITL>(p, cs, cw, i0 + 1, n_pil) : make_float2(0.f, 0.f);
        if (p.sigma) {
            const uint4 r = philox4x32(make_uint4((uint32_t)pr, (uint32_t)f, 1u, 0u), make_uint2(p.seed_lo, p.seed_hi));
            // Box-Muller on the hardware units: v_log_f32, v_sqrt_f32, and v_sin_f32 / v_cos_f32, which take their argument in
            // revolutions (u2 itself) -- the accurate libm forms cost ~10x the instructions and the noise needs none of it
            const float u1 = ((float)r.x + 1.0f) * 2.3283064365386963e-10f, u2 = (float)r.y * 2.3283064365386963e-10f;
            const float u3 = ((float)r.z + 1.0f) * 2.3283064365386963e-10f, u4 = (float)r.w * 2.3283064365386963e-10f;
            const float ra = sg * __builtin_amdgcn_sqrtf(-2.0f * hw_log(u1)), rb = sg * __builtin_amdgcn_sqrtf(-2.0f * hw_log(u3));
            y0.x += ra * __builtin_amdgcn_cosf(u2); y0.y += ra * __builtin_amdgcn_sinf(u2);
            y1.x += rb * __builtin_amdgcn_cosf(u4); y1.y += rb * __builtin_amdgcn_sinf(u4);
        }
        float2 *out = reinterpret_cast<float2 *>(p.pl_out + (size_t)f * 2 * p.pl_frame) + i0;
        if (two && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) *reinterpret_cast<float4 *>(out) = make_float4(y0.x, y0.y, y1.x, y1.y);
        else { out[0] = y0; if (two) out[1] = y1; }
    }
}

// Channel_AWGN::add_noise (DVBS2.cpp:593-613): Y = X + sigma[f] * n, n ~ N(0,1) per real value.
// Philox counter = (pair index, frame, stream 2): independent of the launch geometry.
__global__ void awgn_kernel(const float2 *x, float2 *y, const float *sigma, uint32_t seed_lo, uint32_t seed_hi, long long n_pairs)
{
    const int f = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const uint4 r = philox4x32(make_uint4((uint32_t)i, (uint32_t)f, 2u, (uint32_t)(i >> 32)), make_uint2(seed_lo, seed_hi));
    const float u1 = ((float)r.x + 1.0f) * 2.3283064365386963e-10f, u2 = (float)r.y * 2.3283064365386963e-10f;
    const float rad = sigma[f] * __builtin_amdgcn_sqrtf(-2.0f * hw_log(u1));
    const float sn = __builtin_amdgcn_sinf(u2), cn = __builtin_amdgcn_cosf(u2);
    const float2 v = x[(size_t)f * n_pairs + i];
    y[(size_t)f * n_pairs + i] = make_float2(v.x + rad * cn, v.y + rad * sn);
}
hipError_t awgn_launch(const float *x, float *y, const float *sigma, unsigned long long seed, long long n_pairs, int F, hipStream_t s)
{
    hipLaunchKernelGGL(awgn_kernel, dim3((unsigned)((n_pairs + 255) / 256), F), dim3(256), 0, s, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<float2 *>(y), sigma, (uint32_t)seed, (uint32_t)(seed >> 32), n_pairs);
    return hipGetLastError();
}

hipError_t tx_launch(const TxKParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(tx_src_kernel, dim3(p.n_frames), dim3(256), (size_t)((p.K_bch + 31) / 32) * 4, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tx_bchpar_kernel, dim3((p.n_frames + 64 / TX_BCH_SEG - 1) / (64 / TX_BCH_SEG)), dim3(64), 0, s, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t q_enc = (size_t)((p.N_ldpc - p.K_ldpc) / LDPC_Z);
    const size_t lds = ((size_t)((p.K_ldpc + 31) / 32) + 1 + (q_enc + 1) * ((LDPC_Z + 31) / 32) + q_enc * p.enc_stride) * 4;      // info (+1) | rows | prefix | table
    hipLaunchKernelGGL(tx_ldpc_kernel, dim3(p.n_frames), dim3(LDPC_THREADS), lds, s, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 g(((p.pl_frame + 1) / 2 + 256 * TX_MOD_U - 1) / (256 * TX_MOD_U), p.n_frames), b(256);
    const bool itl = p.itl_cols > 1;
    if (itl && p.itl_cols != p.bps) hipLaunchKernelGGL((tx_mod_kernel<0, false>), g, b, 0, s, p);          // not a DVB-S2 interleaver: general form
    else if (p.bps == 1) hipLaunchKernelGGL((tx_mod_kernel<1, false>), g, b, 0, s, p);
    else if (p.bps == 2 && !itl) hipLaunchKernelGGL((tx_mod_kernel<2, false>), g, b, 0, s, p);
    else if (p.bps == 2) hipLaunchKernelGGL((tx_mod_kernel<2, true>), g, b, 0, s, p);
    else if (p.bps == 3 && !itl) hipLaunchKernelGGL((tx_mod_kernel<3, false>), g, b, 0, s, p);
    else if (p.bps == 3) hipLaunchKernelGGL((tx_mod_kernel<3, true>), g, b, 0, s, p);
    else if (p.bps == 4 && itl) hipLaunchKernelGGL((tx_mod_kernel<4, true>), g, b, 0, s, p);
    else if (p.bps == 5 && itl) hipLaunchKernelGGL((tx_mod_kernel<5, true>), g, b, 0, s, p);
    else hipLaunchKernelGGL((tx_mod_kernel<0, false>), g, b, 0, s, p);
    return hipGetLastError();
}

}  // namespace dvbs2
