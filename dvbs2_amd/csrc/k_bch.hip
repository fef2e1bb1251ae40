// a2 -- DVB-S2 BCH hard-decision decoder (syndromes -> Berlekamp-Massey -> Chien) for gfx950,
// with the BB descrambler (a8) optionally fused into the output store.
//
// Replaces Decoder_BCH_DVBS2<B,R>::_decode_hiho
// (/root/reference src/common/Module/Decoder_BCH_DVBS2/Decoder_BCH_DVBS2.cpp:28-40) and the
// aff3ct Decoder_BCH_std::_decode it wraps; the two reverse_copy calls of the reference
// disappear because the syndromes are evaluated directly in DVB-S2 bit order (position i of
// the frame is the coefficient of x^(N-1-i)).
//
// One frame per 256-thread workgroup.  The frame is packed to a bit image in LDS (N/32 words);
// the odd syndromes are accumulated per set bit from the alpha^k table (L1/L2 resident,
// 2^(m+1) bytes), even ones are squares.  Almost every frame leaving the LDPC decoder has a
// zero syndrome and stops there (H5); otherwise lane 0 runs Berlekamp-Massey (<= 2t steps) and
// all lanes share the Chien search over the 2^m-1 field positions.
// Integer work: results are bit-exact against oracle/dvbs2_oracle.c (orc_bch_decode).
#include "dvbs2hip_internal.h"
#include <cstdlib>

namespace dvbs2 {

std::string bch_build_plan(BchPlan &pl, int m, const int32_t *prim, int t, int N, int K)
{
    // the syndrome kernel splits remainders into a low byte and m - 8 high bits (k_bch.hip, bch_decode_kernel): m < 8 would shift
    // by a negative amount.  DVB-S2 uses m = 14 (short) and 16 (normal).
    if (m < 8 || m > 16) return "BCH: need 8 <= m <= 16 (DVB-S2: 14 or 16)";
    if (t < 1 || t > 12) return "BCH: need 1 <= t <= 12";
    const int n = (1 << m) - 1;
    if (N > n || K >= N || K <= 0) return "BCH: need 0 < K < N <= 2^m-1";
    if (N - K != m * t) return "BCH: N-K must equal m*t for the DVB-S2 codes";
    pl.m = m; pl.n = n; pl.t = t; pl.N = N; pl.K = K;
    int pm = 0;
    for (int i = 0; i <= m; i++) if (prim[i]) pm |= 1 << i;
    if (!(pm >> m) || !(pm & 1)) return "BCH: primitive polynomial must have degree m and constant term";
    pl.exp_.assign(2 * (size_t)n + 2, 0);
    pl.log_.assign((size_t)n + 1, 0);
    int x = 1;
    for (int i = 0; i < n; i++) {
        if (i > 0 && x == 1) return "BCH: polynomial is not primitive";
        pl.exp_[i] = (uint16_t)x; pl.exp_[i + n] = (uint16_t)x; pl.log_[x] = (uint16_t)i;
        x <<= 1; if (x >> m) x ^= pm;
    }
    pl.exp_[2 * (size_t)n] = 1;

    // ---- syndrome tables.  For odd j the minimal polynomial m_j(x) of alpha^j has degree m (DVB-S2
    // codes), GF(2)[x]/m_j is a field and r(x) -> r(alpha^j) factors through r mod m_j: the frame is
    // reduced modulo m_j byte by byte like a CRC, and only the m-bit remainder is evaluated.
    if (N % 8) return "BCH: N must be a multiple of 8";
    auto mul = [&](int a, int b) { return (a && b) ? (int)pl.exp_[pl.log_[a] + pl.log_[b]] : 0; };
    pl.syn_tab.assign(256 + (size_t)768 * t, 0);
    for (int b = 0; b < 256; b++) { int r = 0; for (int i = 0; i < 8; i++) if (b >> i & 1) r |= 1 << (7 - i); pl.syn_tab[b] = (uint16_t)r; }
    for (int kj = 0; kj < t; kj++) {
        const int j = 2 * kj + 1;
        std::vector<int> mp{1};                        // m_j(x) = prod over the coset of j of (x + alpha^e)
        int e = j % n;
        do {
            const int a = pl.exp_[e];
            mp.push_back(0);
            for (int i = (int)mp.size() - 1; i >= 1; i--) mp[i] = mp[i - 1] ^ mul(mp[i], a);
            mp[0] = mul(mp[0], a);
            e = (int)(((long long)e * 2) % n);
        } while (e != j % n);
        if ((int)mp.size() - 1 != m) return "BCH: minimal polynomial of degree != m (not a DVB-S2 code)";
        uint32_t poly = 0;                             // m_j without its leading term
        for (int i = 0; i < m; i++) { if (mp[i] > 1) return "BCH: minimal polynomial not over GF(2)"; if (mp[i]) poly |= 1u << i; }
        uint16_t *T = &pl.syn_tab[256 + (size_t)768 * kj];
        for (int u = 0; u < 256; u++) {                // (u(x) x^m) mod m_j
            uint32_t r = 0;
            for (int i = 7; i >= 0; i--) {             // feed u's bits, highest degree first, through the LFSR
                const uint32_t fb = ((r >> (m - 1)) & 1u) ^ ((u >> i) & 1u);
                r = (r << 1) & ((1u << m) - 1u);
                if (fb) r ^= poly;
            }
            T[u] = (uint16_t)r;
            // evaluation of a remainder byte at alpha^j: sum_b bit_b alpha^(j (b + 8 half))
            uint32_t lo = 0, hi = 0;
            for (int bb = 0; bb < 8; bb++) if (u >> bb & 1) {
                lo ^= pl.exp_[(int)(((long long)j * bb) % n)];
                if (bb + 8 < m) hi ^= pl.exp_[(int)(((long long)j * (bb + 8)) % n)];
            }
            T[256 + u] = (uint16_t)lo; T[512 + u] = (uint16_t)hi;
        }
    }
    // ---- Chien search tables (behind the syndrome tables): thread d of the kernel visits field positions d, d + 256, d + 512, ..; from one to the next the term
    // sigma_i alpha^(i (n - d)) is multiplied by the constant c_i = alpha^(-256 i), and a multiplication by a constant is two byte look-ups:
    // c_i v = LO_i[v & 255] ^ HI_i[v >> 8].  [i][0 .. 256) = LO_i, [i][256 .. 512) = HI_i, i = 0 .. t.
    const size_t c0 = pl.syn_tab.size();
    pl.syn_tab.resize(c0 + (size_t)512 * (t + 1), 0);
    for (int i = 0; i <= t; i++) {
        const int e = (n - (int)(((long long)256 * i) % n)) % n;       // log c_i
        for (int b = 0; b < 256; b++) {
            pl.syn_tab[c0 + (size_t)512 * i + b] = b ? pl.exp_[(pl.log_[b] + e) % n] : 0;
            const int hb = b << 8;
            pl.syn_tab[c0 + (size_t)512 * i + 256 + b] = (b && hb <= n) ? pl.exp_[(pl.log_[hb] + e) % n] : 0;
        }
    }
    return "";
}

// g(x) = lcm of the minimal polynomials of alpha^1 .. alpha^2t (one per cyclotomic coset),
// as tools::BCH_polynomial_generator builds it (TX_RX_BB/main.cpp:45)
std::vector<uint8_t> bch_generator(const BchPlan &pl)
{
    const int n = pl.n;
    auto mul = [&](int a, int b) { return (a && b) ? (int)pl.exp_[pl.log_[a] + pl.log_[b]] : 0; };
    std::vector<uint8_t> g{1};
    std::vector<char> seen(n, 0);
    for (int j = 1; j <= 2 * pl.t; j++) {
        if (seen[j % n]) continue;
        std::vector<int> mp{1};                        // prod (x + alpha^e) over the coset of j
        int e = j % n;
        do {
            seen[e] = 1;
            const int a = pl.exp_[e];
            mp.push_back(0);
            for (int i = (int)mp.size() - 1; i >= 1; i--) mp[i] = mp[i - 1] ^ mul(mp[i], a);
            mp[0] = mul(mp[0], a);
            e = (int)(((long long)e * 2) % n);
        } while (e != j % n);
        std::vector<uint8_t> ng(g.size() + mp.size() - 1, 0);
        for (size_t a = 0; a < g.size(); a++) if (g[a])
            for (size_t c = 0; c < mp.size(); c++) if (mp[c]) ng[a + c] ^= 1;
        g.swap(ng);
    }
    return g;
}

__device__ __forceinline__ uint32_t mod_n(uint32_t x, int m, uint32_t n)
{
    x = (x & n) + (x >> m);
    x = (x & n) + (x >> m);
    return x >= n ? x - n : x;
}

constexpr int BCH_THREADS = 256;
constexpr int BCH_TMAX = 12;

__global__ void __launch_bounds__(BCH_THREADS)
bch_decode_kernel(const BchKParams p)
{
    extern __shared__ uint32_t words[];             // ceil(N/32) packed received bits, then the syndrome tables
    __shared__ uint32_t S[2 * BCH_TMAX + 2];        // S[1..2t]
    __shared__ int Cs[2 * BCH_TMAX + 4], Bs[2 * BCH_TMAX + 4], Ts[2 * BCH_TMAX + 4];
    __shared__ int s_L, s_status, s_nroots, s_any;
    __shared__ int roots[BCH_TMAX + 4];
    __shared__ uint16_t ctab[(BCH_TMAX + 1) * 512];      // Chien step tables, staged by the first frame of this workgroup that needs them
    bool ctab_ready = false;
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = p.N, K = p.K, t = p.t, m = p.m;
    const uint32_t n = (uint32_t)p.n;
    const int nw = (N + 31) / 32;
    uint16_t *stab = reinterpret_cast<uint16_t *>(words + nw);
    bool stab_ready = false;
    auto stage_tables = [&]() {   // the tables, two entries per load (the entry count 256 + 768 t is even, both sides are 4-byte aligned)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.syn_tab);
        uint32_t *dst = reinterpret_cast<uint32_t *>(stab);
        for (int i = tid; i < (256 + 768 * t) / 2; i += BCH_THREADS) dst[i] = src[i];
        stab_ready = true;
    };
    if (!p.flag) { stage_tables(); __syncthreads(); }

    for (int f = blockIdx.x; f < p.n_frames; f += gridDim.x) {
        if (p.flag) {
            // (round 5) behind an LDPC kernel that has checked r(x) mod g(x) itself (k_ldpc_wg8.hip, `syn_tab`): a frame that is a codeword is left alone -- its
            // information bits and its CWD flag are in place -- and a workgroup that meets no flagged frame does not even stage the tables
            if (!p.flag[f]) continue;                 // (uniform over the workgroup)
            if (!stab_ready) stage_tables();          // (the barrier behind the bit image covers it)
        }
        // ---- 1. bit image
        if (p.flag) {
            // the first K bits: the producer's descrambled information bits, scrambled again; the N - K parity bits: the packed bytes of the frame's last row
            const int32_t *src = p.out_bits + (size_t)f * K;
            const uint32_t *pk = p.in_packed + (size_t)f * nw;
            for (int base = 0; base < nw * 32; base += BCH_THREADS) {
                const int i = base + tid;
                int bit = 0;
                if (i < K) { bit = src[i] & 1; if (p.prbs) bit ^= (int)((p.prbs[i >> 5] >> (i & 31)) & 1u); }
                else if (i < N) bit = (int)((pk[i >> 5] >> (i & 31)) & 1u);
                const unsigned long long mask = __ballot(bit);
                if (lane == 0 && (i >> 5) < nw) words[i >> 5] = (uint32_t)mask;
                if (lane == 32 && (i >> 5) < nw) words[i >> 5] = (uint32_t)(mask >> 32);
            }
        } else if (p.in_packed) {
            const uint32_t *src = p.in_packed + (size_t)f * nw;
            for (int w = tid; w < nw; w += BCH_THREADS) words[w] = src[w];
        } else {
            const int32_t *src = p.in_bits + (size_t)f * N;
            for (int base = 0; base < nw * 32; base += BCH_THREADS) {
                const int i = base + tid;
                const int bit = (i < N) ? (src[i] & 1) : 0;
                const unsigned long long mask = __ballot(bit);
                if (lane == 0 && (i >> 5) < nw) words[i >> 5] = (uint32_t)mask;
                if (lane == 32 && (i >> 5) < nw) words[i >> 5] = (uint32_t)(mask >> 32);
            }
        }
        if (tid < 2 * BCH_TMAX + 2) S[tid] = 0u;
        if (tid == 0) { s_L = 0; s_status = 0; s_nroots = 0; s_any = 0; }
        __syncthreads();

        // ---- 2. odd syndromes S_j = r(alpha^j): each lane reduces its chunk of bytes modulo the
        //      minimal polynomial m_j (table-driven, CRC style, tables in LDS), evaluates the m-bit
        //      remainder at alpha^j and shifts it to the chunk's place: x alpha^(8 j bytes_after).
        {
            const int nbytes = N / 8;
            const int L = (nbytes + BCH_THREADS - 1) / BCH_THREADS;          // bytes per lane
            const int b0 = tid * L, b1 = min(b0 + L, nbytes);
            uint32_t rem[BCH_TMAX];
#pragma unroll
            for (int kk = 0; kk < BCH_TMAX; kk++) rem[kk] = 0u;
            const uint32_t lowmask = (1u << (m - 8)) - 1u;
            for (int by = b0; by < b1; by++) {
                const uint32_t raw = (words[by >> 2] >> ((by & 3) * 8)) & 0xFFu;
                const uint32_t B = stab[raw];                                 // first bit = highest degree
#pragma unroll
                for (int kk = 0; kk < BCH_TMAX; kk++)
                    if (kk < t) rem[kk] = (uint32_t)stab[256 + 768 * kk + (rem[kk] >> (m - 8))] ^ ((rem[kk] & lowmask) << 8) ^ B;
            }
            const uint32_t after = (uint32_t)(nbytes - b1);                   // bytes behind this chunk
#pragma unroll
            for (int kk = 0; kk < BCH_TMAX; kk++)
                if (kk < t) {
                    uint32_t v = 0u;
                    if (b1 > b0) {
                        v = (uint32_t)stab[256 + 768 * kk + 256 + (rem[kk] & 0xFFu)] ^ (uint32_t)stab[256 + 768 * kk + 512 + (rem[kk] >> 8)];
                        if (v) v = p.exp_[(uint32_t)p.log_[v] + mod_n(mod_n((uint32_t)(2 * kk + 1) * 8u, m, n) * after, m, n)];
                    }
                    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
                    if (lane == 0 && v) atomicXor(&S[2 * kk + 1], v);
                }
        }
        __syncthreads();
        if (tid == 0) {
            int any = 0;
            for (int j = 2; j <= 2 * t; j += 2) {          // S_2k = S_k^2
                const uint32_t h = S[j / 2];
                S[j] = h ? p.exp_[2u * p.log_[h]] : 0u;
            }
            for (int j = 1; j <= 2 * t; j++) any |= (int)S[j];
            s_any = any;
        }
        __syncthreads();

        if (s_any) {
            // ---- 3. Berlekamp-Massey (<= 2t steps; same recurrence as the oracle).  (round 4) By the first WAVE instead of the first lane: lane i holds C[i] and B[i]; a step's
            // discrepancy is one product per lane (three table look-ups, all lanes' in flight together) and an XOR over the wave, its update one product per lane with B taken
            // from lane i - m.  On one lane the look-ups of a step -- global memory, each behind the one before -- made a frame that does not decode cost 0.19 ms (t = 12) / 0.36 ms
            // (t = 8, GF(2^16)), and ONE such frame in a batch is what the launch then takes; the field operations and their order per coefficient are unchanged.
            if (tid < 64) {
                const uint16_t *ex = p.exp_, *lg = p.log_;
                constexpr int lim = 2 * BCH_TMAX + 4;
                int Ci = lane == 0 ? 1 : 0, Bi = lane == 0 ? 1 : 0;          // (lanes >= lim stay 0)
                int L = 0, mm = 1, bb = 1;
                for (int k = 0; k < 2 * t; k++) {
                    int term = 0;
                    if (lane >= 1 && lane <= L && Ci) { const int sv = (int)S[k + 1 - lane]; if (sv) term = ex[lg[Ci] + lg[sv]]; }
                    for (int o = 32; o > 0; o >>= 1) term ^= __shfl_xor(term, o);
                    const int d = (int)S[k + 1] ^ term;
                    if (d == 0) { mm++; continue; }
                    const int lcoef = (int)lg[d] + (int)n - (int)lg[bb];       // log(d / b)
                    const int Bsrc = __shfl(Bi, lane - mm);                   // B[i - m] (lanes below m: no term)
                    int upd = 0;
                    if (lane >= mm && lane < lim && Bsrc) upd = ex[mod_n((uint32_t)(lcoef + lg[Bsrc]), m, n)];
                    if (2 * L <= k) {
                        const int Told = Ci;
                        Ci ^= upd;
                        L = k + 1 - L;
                        Bi = Told;
                        bb = d; mm = 1;
                    } else {
                        Ci ^= upd;
                        mm++;
                    }
                }
                if (lane < lim) Cs[lane] = Ci;
                if (lane == 0) { s_L = L; if (L > t) s_status = 1; }
            }
            __syncthreads();
            const int L = s_L;
            if (!s_status) {
                // ---- 4. Chien search over the whole field: sigma(alpha^-d) == 0 <=> error at degree d.  (round 4) A thread's positions d, d + 256, .. differ by a constant
                // factor per term, so one exp look-up per term STARTS the thread and every further position is two LDS byte look-ups per term (bch_build_plan's c_i tables)
                // instead of a multiply-modulo and a global look-up: the search over GF(2^16) was 0.3 ms of a frame that does not decode.
                if (!ctab_ready) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.syn_tab + 256 + 768 * t);
                    uint32_t *dst = reinterpret_cast<uint32_t *>(ctab);
                    for (int i = tid; i < (t + 1) * 256; i += BCH_THREADS) dst[i] = src[i];
                    ctab_ready = true;
                    __syncthreads();
                }
                uint32_t term[BCH_TMAX + 1];
                {
                    const uint32_t nd = n - (uint32_t)tid;        // tid < 256 <= n
#pragma unroll
                    for (int i = 0; i <= BCH_TMAX; i++)
                        term[i] = (i <= L && Cs[i]) ? (uint32_t)p.exp_[(uint32_t)p.log_[Cs[i]] + mod_n((uint32_t)i * nd, m, n)] : 0u;
                }
                for (uint32_t d = tid; d < n; d += BCH_THREADS) {
                    uint32_t v = 0u;
#pragma unroll
                    for (int i = 0; i <= BCH_TMAX; i++) v ^= term[i];
                    if (v == 0u) {
                        const int slot = atomicAdd(&s_nroots, 1);
                        if (slot < BCH_TMAX + 4) roots[slot] = (int)d;
                    }
#pragma unroll
                    for (int i = 1; i <= BCH_TMAX; i++)
                        if (i <= t) term[i] = (uint32_t)ctab[512 * i + (term[i] & 0xFFu)] ^ (uint32_t)ctab[512 * i + 256 + (term[i] >> 8)];
                }
                __syncthreads();
                if (s_nroots == L) {
                    if (tid < L) {
                        const int pos = N - 1 - roots[tid];
                        if (pos >= 0 && pos < K) {
                            atomicXor(&words[pos >> 5], 1u << (pos & 31));
                            if (p.patch_only) p.out_bits[(size_t)f * K + pos] ^= 1;          // the producer wrote the uncorrected (descrambled) bit
                        }
                    }
                } else if (tid == 0) s_status = 1;
                __syncthreads();
            }
        }

        // ---- 5. output the K systematic bits (optionally BB-descrambled, Scrambler_BB.hxx:51-72)
        int32_t *out = p.out_bits + (size_t)f * K;
        if (p.patch_only) {
            // fused chain: the LDPC kernel has written the K descrambled bits of this frame already (k_ldpc_wg8.hip, `info_out`)
        } else if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            // four bits per lane: one 16-byte non-temporal store (the socket is written once and read by another kernel)
            typedef int32_t bch_i4 __attribute__((ext_vector_type(4)));
#pragma unroll 8
            for (int k = 4 * tid; k < K; k += 4 * BCH_THREADS) {          // unrolled: the PRBS words (global, cache-resident) of 8 iterations in flight
                uint32_t b = words[k >> 5];
                if (p.prbs) b ^= p.prbs[k >> 5];
                b >>= (k & 31);
                bch_i4 v;
                v.x = (int32_t)(b & 1u); v.y = (int32_t)((b >> 1) & 1u); v.z = (int32_t)((b >> 2) & 1u); v.w = (int32_t)((b >> 3) & 1u);
                __builtin_nontemporal_store(v, reinterpret_cast<bch_i4 *>(out + k));
            }
        } else
        for (int k = tid; k < K; k += BCH_THREADS) {
            uint32_t b = (words[k >> 5] >> (k & 31)) & 1u;
            if (p.prbs) b ^= (p.prbs[k >> 5] >> (k & 31)) & 1u;
            out[k] = (int32_t)b;
        }
        if (tid == 0 && p.cwd) p.cwd[f] = s_status ? 0 : 1;
        __syncthreads();
    }
}

hipError_t bch_launch(const BchPlan &pl, BchKParams p, hipStream_t s)
{
    p.exp_ = pl.d_exp; p.log_ = pl.d_log; p.syn_tab = pl.d_syn_tab;
    p.N = pl.N; p.K = pl.K; p.m = pl.m; p.n = pl.n; p.t = pl.t;
    const size_t lds = (size_t)((pl.N + 31) / 32) * 4 + (256 + (size_t)768 * pl.t) * 2;
    // persistent grid (about the number of workgroups the chip holds at once): the tables are staged into LDS once per
    // workgroup, not once per frame
    static const int cap = [] { const char *e = getenv("DVBS2HIP_BCH_GRID"); return e ? atoi(e) : 2048; }();
    const int grid = p.n_frames < cap ? p.n_frames : cap;
    hipLaunchKernelGGL(bch_decode_kernel, dim3(grid), dim3(BCH_THREADS), lds, s, p);
    return hipGetLastError();
}

}  // namespace dvbs2
