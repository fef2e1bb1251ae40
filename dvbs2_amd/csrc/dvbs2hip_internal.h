// Internal declarations shared by the HIP translation units of libdvbs2hip.so.
// Nothing here is part of the ABI (include/dvbs2hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/dvbs2hip.h"

namespace dvbs2 {

// e^x and ln x on the hardware transcendental units, ONE v_exp_f32 / v_log_f32 plus a multiply each.  HIP's __expf / __logf
// wrap the same instructions in a range check and a rescale for denormal results (8 instructions instead of 2); the
// arguments here (exp of a non-positive number, log of a sum >= 1 or of a sum of such exponentials) do not need that:
// a result below the smallest normal number flushes to zero, which is what the formulas treat it as.
#if defined(__HIPCC__)
__device__ __forceinline__ float hw_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float hw_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
#endif

// ---------------------------------------------------------------- LDPC (a1)
// One slot of a QC layer = one circulant: variable = group element (t - t0) mod 360.
// Packed in ONE dword so a whole layer (<= 27 entries) is preloaded into SGPRs by a few wide
// scalar loads:  bits 0..8 circulant shift t0 | bits 9..16 slot of the bit-group in its store
// (word offset = 360 * slot) | bit 17 store is LDS | bit 18 edge absent for t == 0 |
// bits 19..20 conflict level | bit 21 padding entry (no edge).
typedef uint32_t LdpcEntry;
enum { LE_T0_MASK = 0x1FF, LE_SLOT_SHIFT = 9, LE_SLOT_MASK = 0xFF, LE_LDS = 1u << 17, LE_MASK0 = 1u << 18,
       LE_LVL_SHIFT = 19, LE_LVL_MASK = 3, LE_NULL = 1u << 21 };

struct LdpcGroup {    // where the 360 posteriors of one bit-group live
    uint32_t base;
    uint32_t lds;     // 1 = LDS, 0 = per-frame global workspace
};

constexpr int LDPC_Z = 360;          // DVB-S2 parallelism factor
constexpr int TX_BCH_SEG = 16;       // segments (lanes) per frame of the TX BCH encoder (k_tx.hip, host-made shift table): 32 / 64 help 512-frame calls (61 -> 46 / 41 us) and cost 4096-frame ones (0.754 -> 0.768 / 0.812 ms per TX)
constexpr int LDPC_THREADS = 384;    // 6 wavefronts, 360 active lanes
constexpr int LDPC_MAX_SLOTS = 27;   // 27 sign bits + 5 index bits = one packed dword

struct LdpcKParams {
    const float *llr;        // [F][N]
    int32_t *bits;           // [F][K]            (may be null)
    uint32_t *packed;        // [F][K/32 words]   (may be null) packed hard decisions, bit i of word w = info bit 32 w + i
    int8_t *cwd;             // [F]               (may be null)
    float *post;             // [F][N] natural    (may be null)
    int32_t *ites;           // [F]               (may be null)
    // fused chain (k_ldpc_wg8.hip): the first K_info hard decisions XOR the BB descrambling sequence, one int32 per bit, straight
    // into the chain's output socket -- what the BCH stage would write for a frame it does not have to correct
    int32_t *info_out;       // [F][K_info]       (may be null)
    const uint32_t *info_prbs;  // PRBS by (row g, wave w): 64-bit word [g * 6 + w], bit l = PRBS bit 360 g + 64 w + l (zero from K_info on); wave-uniform scalars
    int32_t K_info;
    // (round 5) fused chain, BCH verification inside the LDPC kernel: a frame is a BCH codeword iff r(x) mod g(x) = 0.  Bit t of information row g is the coefficient
    // of x^(360 (G - 1 - g)) x^(359 - t): each lane XORs the row's factor A_g into its accumulator for every set hard decision it outputs, the wave applies the lanes'
    // factors once per frame (k_ldpc_wg8.hip, fold_syn).  The BCH stage then runs only over the flagged frames.
    const uint32_t *syn_tab;    // [syn_rows][8]: per row in STORAGE order (the order k_ldpc_wg8.hip emits the rows in: LDS rows, global rows, register slots) {byte offset 1440 g of the row in the int32
                                // socket (0x7FFFF000: an empty register slot), 1 = the last information row, A_g[0 .. 5] = x^(360 (n_info - 1 - g)) mod g(x)}, then [LDPC_SYN_RED][syn_words]: x^k mod g(x),
                                // little-endian 32-bit words; or null (no verification)
    const uint32_t *info_prbs_s;   // [syn_rows][LDPC_AT_LANES]: bit k of entry [ks][t] = the BB descrambler's bit of information bit 360 g + t, g the row emitted at storage position ks + k (k < 16)
    int32_t syn_words;          // 4 (deg g <= 128) or 6 (<= 192)
    int32_t syn_rows;           // rows the kernel emits (information rows + empty register slots)
    uint8_t *bch_flag;          // [F]: 1 = remainder not zero (the BCH stage decodes this frame), 0 = codeword
    int8_t *cwd_bch;            // [F] or null: the BCH stage's CWD flag, written (1) for a frame whose remainder is zero
    float *gwork;            // [grid][gwork_words] per-workgroup global workspace
    const LdpcEntry *entries;  // [q][deg_max_padded]
    const int32_t *layer_deg;  // [q]
    const int32_t *layer_lvl;  // [q] max conflict level in the layer
    const LdpcGroup *groups;   // [n_groups] info groups then parity groups
    int32_t N, K, M, q, n_info, n_groups;
    int32_t ent_stride;      // entries per layer in `entries`
    int32_t lds_post_words;  // LDS words used by posteriors
    int32_t glb_post_words;  // global words per frame used by posteriors
    int32_t gwork_words;     // global words per frame in total (posteriors + c2v state)
    int32_t n_frames, n_ite, early_stop;
    float alpha;
    const uint32_t *order;        // queue position -> frame index (n_frames entries), or null: frames in index order (k_ldpc_wg8.hip; built by frame_order_launch)
    float spa_cap;                // sum-product, exact check node: |c->v| is clipped to this (LDPC_SPA_CAP: the reference's results; +inf: no clip)
    int32_t inf_row;           // fast path: byte offset of the +inf row (padded layers) or -1
    uint32_t *cu_ctr;          // 8-wave workgroups: per-CU arrival counter (zeroed before the launch) or null
    struct {                   // k_ldpc_wg8.hip
        const uint32_t *tab;   // [q][LDPC_FAST_STRIDE]: byte shift | byte offset of the bit-group row << 11 | LDS flag << 29; modes 4 / 5: then [q][NR] the idle waves' swaps (LDS position or 0xFF)
        const uint32_t *atab;  // per-lane address table (LdpcPlan::w8_atab) or null
        const uint32_t *rows;  // bit-group of LDS row l (nl of them), then of global row l (ng), then where the q parity groups live; modes 4 / 5: then the bit-group in register slot k (NR)
        uint32_t st_base;      // byte offset of the packed c->v state in the workgroup's global slot
        uint32_t lds_junk;     // byte offset of the write-only LDS row (the +inf row of padded codes follows it)
        int32_t lds_bytes, pad;
        int32_t nl_info, nl, ng_info, ng;
    } w8;
};

struct LdpcPlan {             // host-side description, built once per handle
    int N = 0, K = 0, M = 0, q = 0, n_info = 0, n_groups = 0, E = 0;
    int deg_max = 0, ent_stride = 0;
    int lds_groups = 0;
    bool c2v_lds = false, hybrid = false;
    int lds_post_words = 0, glb_post_words = 0, gwork_words = 0;
    size_t lds_bytes = 0;
    int grid_max = 1;         // persistent grid: resident workgroups on the device
    int n_cus = 256;          // compute units of the device (k_ldpc_nat.hip: the natural order's workgroup shape by batch size)
    std::vector<LdpcEntry> entries;
    std::vector<int32_t> layer_deg, layer_lvl;
    std::vector<LdpcGroup> groups;
    // device copies
    LdpcEntry *d_entries = nullptr;
    int32_t *d_layer_deg = nullptr, *d_layer_lvl = nullptr;
    LdpcGroup *d_groups = nullptr;
    // regular-code fast path (k_ldpc_wg8.hip): every layer has exactly fast_deg slots
    bool fast = false;
    bool spa = false;             // sum-product check node: per-edge fp32 messages instead of the packed min-sum state
    int spa_rule = 0;             // 0: not sum-product, 1: exact (complement-product domain), 2: AFF3CT's saturating tanh-product form, bit for bit the oracle's ORC_SPA_TANH, 3: the exact rule with every message clipped to LDPC_SPA_CAP (same kernels as 1, LdpcKParams::spa_cap)
    int fast_deg = 0;             // slots per layer in the unrolled kernel (11, 13 or 27)
    bool fast_pad = false;        // layers padded with NULL slots (irregular code)
    int fast_inf_row = -1;        // byte offset of the +inf row the NULL slots read, or -1
    int fast_mode = 0;            // posterior image -- 0: in LDS, 1: in the workgroup's global slot, 3: static hybrid (k_ldpc.hip plan), 4 / 5: static hybrid + 32 / 39 rows parked in the idle waves' registers
    std::vector<uint32_t> nat_tab, nat_haz;   // k_ldpc_nat.hip: natural-row-order tables
    uint32_t *d_nat_tab = nullptr, *d_nat_haz = nullptr;
    bool fast_cu1 = false;        // mode 6: one frame per CU, the whole image on chip (k_ldpc_cu1.hip); fast_wg8 is set too (same tables)
    int cu1_pairs = 0;            // mode 6: pairs of rows that share an LDS position and a register slot of the row-keeping waves
    bool fast_wg8 = false;        // one frame per 8-wave workgroup, SIMD-aware roles, two independent workgroups per CU (k_ldpc_wg8.hip)
    bool w8_dups_in_lds = false;  // static hybrid: every bit-group with two edges in one layer is LDS-resident
    std::vector<uint32_t> w8_tab, w8_rows;
    uint32_t *d_w8_tab = nullptr, *d_w8_rows = nullptr;
    // (round 4) per-lane address table of the min-sum layer: [q][ceil(deg / 4)][LDPC_AT_LANES][4] dwords -- for slot j of layer r and check t the LDS byte address of the
    // posterior (LDS slot) or its rotated byte offset inside the row (global slot): what the layer loop formed with 3-4 vector instructions per slot and iteration
    std::vector<uint32_t> w8_atab;
    uint32_t *d_w8_atab = nullptr;
    uint32_t w8_st_base = 0, w8_lds_junk = 0;
    int w8_lds_bytes = 0, w8_gwork_words = 0, w8_nl_info = 0, w8_nl = 0, w8_ng_info = 0, w8_ng = 0;
    int w8_park_moves = 0;        // mode 4: row moves between LDS and the idle waves' registers per iteration
    uint32_t *d_cu_ctr = nullptr; // [LDPC_CU_CTR_WORDS]
};
constexpr int LDPC_PROF_WORDS = 1024 * 64;   // development aid (LDPC_PHASE_PROF builds): per-wave phase timers
constexpr int LDPC_CU_CTR_WORDS = 4096 + 64;   // [0, 4096): arrivals per CU, key = XCC_ID << 8 | SE_ID << 5 | SH_ID << 4 | CU_ID; [4096]: frames handed out (work queue)
constexpr int LDPC_FRAME_CTR = 4096;
#ifndef LDPC_ATAB_HYB
#define LDPC_ATAB_HYB 0          // the min-sum layer's per-lane address table for the LDS slots of the hybrid images too (k_ldpc_wg8.hip W8_ATAB_HYB: measured and left off); 0: the plan builds
#endif                           // and uploads the table for the LDS-only image alone (0.9-6 MB per handle that no normal-frame kernel reads)
#ifndef LDPC_SPA_AT16
#define LDPC_SPA_AT16 1      // sum-product layer, LDS-only image: the slots' LDS addresses from a per-lane table of 16-bit entries (k_ldpc.hip plan, k_ldpc_wg8.hip W8_SPA_AT16) ...
#endif
#ifndef LDPC_SPA_AT16_MAXDEG
#define LDPC_SPA_AT16_MAXDEG 13      // ... for the 11- and 13-slot codes only (QPSK-S 3/5: 14.33 -> 13.47 ms per 16384 frames).  The 27-slot layer runs at the 128-register budget of two workgroups
#endif                               // per CU: the 14 registers that carry the table do not exist there (23 / 19 / 8 spilled registers with suffix values every 3rd / 4th / 6th slot: QPSK-S 8/9
                                     // 9.63 -> 13.0 / 11.9 / 11.2 ms, docs/negative_results.md)
constexpr int LDPC_AT_LANES = 384;      // lanes per row of the address table (6 waves; lanes 360 .. 383 hold the junk row / an offset that is dropped)
constexpr int LDPC_FAST_STRIDE = 64;   // dwords per layer: 27 entries | prim mask | conflict info | conflict entries 0, 1 | slots with a duplicate edge | 16 conf entries | 16 conf meta
                                       // (sum-product plans, at most LDPC_SPA_MAXC conflict entries: dwords LDPC_TANH_ORDER .. +4 = the slots in the ORACLE's edge order, 5 bits each, 6 per dword)
constexpr int LDPC_TANH_ORDER = 56;
constexpr float LDPC_SPA_CAP = 16.6355324f;      // 2 atanh(1 - FLT_EPSILON): where the messages of AFF3CT's tanh-product rule stop (spa_rule 2 by construction, 3 by a clip)
constexpr int LDPC_FAST_MAXC = 16;
// modes 4 / 5 (k_ldpc_wg8.hip): bit-group rows parked in the registers of a workgroup's two idle waves (3 VGPRs per row and lane) and LDS slots per
// layer (the static hybrid without parked rows, mode 3, has 9).  Mode 5 (min-sum kernel only: the sum-product kernel has no registers for it) parks 39.
__host__ __device__ __forceinline__ constexpr int ldpc_park_nr(int mode) { return mode == 5 ? 39 : 32; }
__host__ __device__ __forceinline__ constexpr int ldpc_park_nl(int mode) { return mode == 5 ? 15 : 14; }
// k_ldpc_wg8.hip: duplicate edges (the slots whose stores are redirected to the junk row) sit in the first ldpc_w8_kd(deg) slots of a layer; from
// there on a slot's store goes where its load came from (one address per LDS slot instead of two).  The padded 13-slot form (32APSK-S 3/4: checks of degree
// 9 .. 13) has its NULL slots there too (at most 4 + 2 duplicates).
__host__ __device__ __forceinline__ constexpr int ldpc_w8_kd(int deg) { return deg == 27 ? 6 : deg == 11 ? 3 : deg == 13 ? 6 : deg; }
// mode 6 (k_ldpc_cu1.hip): one frame per 16-wave workgroup = per CU.  Two lanes per check: the first half-check's lanes take slots 0 .. LDPC_CU1_HA-1 (the duplicate
// edges are among them), the second one's the rest (p_c and p_{c-1} last); they exchange {min1 | parity, min2} through 8 bytes per half-check of LDS.  Four
// row-keeping waves in two groups of two, ldpc_cu1_nrg() rows (3 VGPRs per row and lane) each.
#ifndef LDPC_CU1_HA_V
#define LDPC_CU1_HA_V 14
#endif
constexpr int LDPC_CU1_HA = LDPC_CU1_HA_V;
constexpr int LDPC_CU1_XCHG_BYTES = 2 * LDPC_Z * 8;
__host__ __device__ __forceinline__ constexpr int ldpc_cu1_nrg() { return 36; }
#ifndef LDPC_CU1_DEFAULT
#define LDPC_CU1_DEFAULT 0
#endif
#ifndef LDPC_CU1_SPA_DEFAULT      // the sum-product decoder on normal frames: mode 6 (one frame per CU, two lanes per check, messages inside the Infinity Cache) as the default
#define LDPC_CU1_SPA_DEFAULT 1
#endif
constexpr int LDPC_W8_MISC_BYTES = 96;  // k_ldpc_wg8.hip: words behind the image -- [0..7] SIMD of wave w, [8] first / second workgroup of the CU, [9] next frame, [10..12] vote words, [16..21] BCH remainder of the frame
constexpr int LDPC_SYN_RED = 640;       // entries x^k mod g(x), k = 0 .. 639, of the BCH verification's reduction table (k <= 64 * 5 + 32 * 8 + 63)
constexpr int LDPC_SPA_MAXC = 6;       // SPA: duplicate edges per layer whose old messages a lane keeps in registers (the DVB-S2 codes have at most 6)
constexpr int NAT_AHEAD = 24;           // k_ldpc_nat.hip, lanes-per-frame form: checks whose loads are in flight ahead of the one being computed (8 lanes per frame; 12 with 4 lanes per frame)
constexpr int NAT_HAZ_WINDOW = NAT_AHEAD + 1;
hipError_t ldpc_nat_launch(const LdpcPlan &pl, const LdpcKParams &kp, float *work, hipStream_t s);
size_t ldpc_nat_group_words(const LdpcPlan &pl);
hipError_t ldpc_wg8_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s);
hipError_t ldpc_cu1_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s);
int ldpc_wg8_blocks_per_cu(const LdpcPlan &pl);

// builds the layer tables; returns empty string on success, else the error text
std::string ldpc_build_plan(LdpcPlan &pl, int N, int K, int n_rows, const int32_t *row_ptr,
                            const int32_t *addr, int lds_groups_req, size_t lds_limit_bytes, int spa_rule = 0, bool small_batch = false);      // small_batch: the handle never sees more frames than CUs      // spa_rule: LdpcPlan::spa_rule
hipError_t ldpc_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s);
int ldpc_blocks_per_cu(const LdpcPlan &pl);

// ---------------------------------------------------------------- BCH (a2)
struct BchPlan {
    int m = 0, n = 0, t = 0, N = 0, K = 0;
    std::vector<uint16_t> exp_, log_;
    uint16_t *d_exp = nullptr, *d_log = nullptr;   // exp has 2n entries (no modulo on sums)
    uint32_t *d_prbs = nullptr;                    // BB scrambler sequence, packed, K bits
    uint32_t *d_prbs_rw = nullptr;                 // the same by (bit-group row, wave) for the LDPC kernel's fused output (LdpcKParams::info_prbs)
    // syndrome tables, one set per odd j = 1, 3, .., 2t-1 (see k_bch.hip):
    //   [0 .. 256)            bit-reversed byte
    //   [256 + 768 k ..)      for the k-th odd j: 256 x (u(x) x^m mod m_j(x)) | 256 x eval(low byte) | 256 x eval(high byte)
    std::vector<uint16_t> syn_tab;
    uint16_t *d_syn_tab = nullptr;
};
struct BchKParams {
    const int32_t *in_bits;     // [F][N] or null
    const uint32_t *in_packed;  // [F][ceil(N/32)] or null
    // (round 5) fused chain behind an LDPC kernel that has verified the frames itself: only frames with flag[f] != 0 are decoded; their first K bits are
    // out_bits (descrambled by the producer: re-scrambled here with `prbs`), the N - K parity bits come from in_packed
    const uint8_t *flag;        // [F] or null (every frame is decoded)
    int32_t *out_bits;          // [F][K]
    int8_t *cwd;                // [F] or null
    const uint16_t *exp_, *log_, *syn_tab;
    const uint32_t *prbs;       // packed K-bit BB descrambling sequence or null (no descramble)
    int32_t patch_only;         // out_bits already holds the (descrambled) uncorrected bits: only flip what the decoder corrects
    int32_t N, K, m, n, t, n_frames;
};
std::string bch_build_plan(BchPlan &pl, int m, const int32_t *prim, int t, int N, int K);
hipError_t bch_launch(const BchPlan &pl, BchKParams p, hipStream_t s);

// ---------------------------------------------------------------- front end (a3, a4, a6, a7)
struct FrontKParams {
    const float *in;        // pl frames [F][2*pl_frame] or xfec frames [F][2*n_sym]
    const float *const *src;   // (round 5) per-frame start of the PL frame (8-byte aligned) instead of in + f * 2 * pl_frame: the located form of the frame synchronizer; or null
    const float *sigma_in;  // [F] or null
    float *llr;             // [F][N_ldpc]
    float *est;             // [F][3] sigma, ebn0, esn0 (may be null)
    const float *cstl;      // normalised constellation, 2^bps points
    const uint8_t *pl_seq;  // PL scrambling sequence R(i), 66420 entries
    int32_t n_sym, pl_frame, bps, itl_cols, itl_order, n_frames;
    int32_t slice_chunk;    // front_kernel: symbols per workgroup when a frame is dealt to several (blockIdx.y), 0 = the whole frame (set by the launcher: small batches)
    float code_rate;
    // separable 2-bit constellation (plan time): L_b = c (y[sep_ax[b]] * sep_g[b] + sep_h[b]); sep = 0: general demapper
    int32_t sep, sep_ax[2];
    float sep_g[2], sep_h[2];
};
size_t ldpc_lat_lds_bytes(const LdpcPlan &pl);                                    // k_ldpc_lat.hip: small batches of short frames, two lanes per check (0: not applicable)
hipError_t ldpc_lat_launch(const LdpcPlan &pl, LdpcKParams p, hipStream_t s);
hipError_t frame_order_launch(const float *llr, float *metric, uint32_t *order, int F, int N, hipStream_t s);      // k_ldpc.hip: the work queue's order for launches with the stopping rule (noisiest frames first)
hipError_t front_rx_launch(FrontKParams p, hipStream_t s);                     // a7+a6+a3+a4 fused, in = pl frames
hipError_t demod_launch(FrontKParams p, bool deinterleave, hipStream_t s);     // a3 (+a4), in = xfec frames, sigma_in required
hipError_t deinterleave_launch(const float *itl, float *nat, int N, int cols, int order, int F, hipStream_t s);
hipError_t estimate_launch(const float *x, float *sig, float *ebn0, float *esn0, int n_sym, float code_rate,
                           int bps, int F, hipStream_t s);
hipError_t pl_descramble_launch(const float *in, float *out, const uint8_t *seq, int pl_frame, int F, hipStream_t s);
hipError_t agc_launch(const float *X, float *Z, int n_cplx, float output_energy, int F, hipStream_t s);      // k_agc.hip
hipError_t nco_launch(const float *X, float *Z, float omega, uint32_t n0, long long total, float *FRQ, float *PHS, float frq, int F, hipStream_t s);    // k_agc.hip
hipError_t remove_plh_launch(const float *in, float *out, int n_sym, int pl_frame, int F, hipStream_t s);
hipError_t bb_descramble_launch(const int32_t *in, int32_t *out, const uint32_t *prbs, int K, int F, hipStream_t s);
hipError_t monitor_launch(const int32_t *U, const int32_t *V, unsigned long long *ctr, int K, int F, hipStream_t s);
hipError_t monitor2_launch(const int32_t *U, const int32_t *V, unsigned long long *ctr, int32_t *be_tmp, long long *FRA, int32_t *BE, int32_t *FE, float *BER,
                           float *FER, int K, int F, hipStream_t s);

// ---------------------------------------------------------------- TX mirror + AWGN (N1)
struct TxKParams {
    const int32_t *info_in;     // [F][K_bch] or null: random payload from (seed, frame)
    int32_t *info_out;          // [F][K_bch] or null
    const float *sigma;         // [F] or null: no noise
    float *pl_out;              // [F][2*pl_frame]
    uint32_t *bch_cw;           // [F][ceil(K_ldpc/32)] work
    uint32_t *ldpc_cw;          // [F][ceil(N_ldpc/32)] work
    const uint32_t *prbs;       // BB scrambling sequence, packed
    const uint32_t *enc_tab;    // [q][enc_stride]: t0 | group << 9
    const int32_t *enc_deg;     // [q]
    const float *cstl;          // normalised constellation
    const float *plh;           // 180 floats: PLHEADER
    const uint8_t *pl_seq;
    const unsigned long long *bch_tab;  // [256][3]: (u(x) x^r) mod g(x), byte-wise systematic encoder
    const unsigned long long *bch_shift;   // [16 segments][r][3]: x^(b + 8 * bytes behind the segment) mod g (segmented division)
    uint32_t seed_lo, seed_hi;
    int32_t K_bch, K_ldpc, N_ldpc, bps, itl_cols, itl_order, n_sym, pl_frame, enc_stride, n_frames;
};
hipError_t tx_launch(const TxKParams &p, hipStream_t s);
// generator polynomial of the t-error-correcting BCH code over GF(2^m): g[i] = coeff of x^i
std::vector<uint8_t> bch_generator(const BchPlan &pl);

// ---------------------------------------------------------------- FIR (a5)
hipError_t fir_launch(const float *x, float *y, const float *hist_in, float *hist_out, const float *taps_rev, const uint16_t *afrag,
                      int T, long long n_total, hipStream_t s);
// matrix-core form (k_fir_mfma.hip): bf16 x 3 split operands, fp32 accumulation
std::vector<uint16_t> fir_mfma_afrag(const float *taps_rev, int T);
bool fir_mfma_usable(const float *x, const float *y, int T, long long n_total);
hipError_t fir_mfma_launch(const float *x, float *y, const float *hist_in, float *hist_out, const uint16_t *afrag, int T, long long n_total, hipStream_t s);

hipError_t upfir_launch(const float *x, float *y, const float *hist_in, float *hist_out, const float *taps, const uint16_t *afrag2, int T, int osf,
                        long long n_in, hipStream_t s);
std::vector<uint16_t> upfir_mfma_afrag(const float *taps, int T);
bool upfir_mfma_usable(const float *x, const float *y, int T, int osf, long long n_in);
hipError_t upfir_mfma_launch(const float *x, float *y, const float *hist_in, float *hist_out, const uint16_t *afrag2, int T, long long n_in, hipStream_t s);
hipError_t decimate_launch(const float *x, float *y, long long n_out, int osf, long long offset, long long n_in, hipStream_t s);
hipError_t awgn_launch(const float *x, float *y, const float *sigma, unsigned long long seed, long long n_pairs, int F, hipStream_t s);

// ---------------------------------------------------------------- frame synchronizer (N4, k_sync.hip)
// per-frame results of the frame synchronizer's arg max (sync_finalize_kernel): the sockets, the delay line's table, the handle's state
struct SyncTail {
    unsigned long long *keys;        // F x ceil(n / 64) words of device scratch
    int32_t *delay; float *metric; int32_t *flag; float trigger;
    int32_t *Dtab; float *last_metric;
    float *seg;                      // 2 x ceil(max_frames / SYNC_SUB) x n floats of device scratch: per sub-segment of the call's frames {alpha^len, its own average from 0} (k_sync.hip)
};
constexpr int SYNC_MAX_SEG = 8, SYNC_SUB = 64;
hipError_t sync_corr_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, float *cor_sof, float *cor_plsc, long long n_total, hipStream_t s);
std::vector<uint16_t> sync_mfma_frag(const float *sof25, const float *plsc64);       // k_sync_mfma.hip: band fragments of the two correlators
bool sync_mfma_usable(const float *x, const void *frag);
hipError_t sync_corr_mfma_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, float *cor_sof, float *cor_plsc, long long n_total, hipStream_t s);
hipError_t sync_corr_m_mfma_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, const float *sofh_in, float *sofh_out, float *corr,
                                   long long n_total, hipStream_t s);
std::vector<uint16_t> sync_frag_default();                                            // k_sync.hip: the fragments of conj_SOF / conj_PLSC
hipError_t sync_corr_metric_launch(const float *x, const float *xh_in, float *xh_out, const uint16_t *frag, const float *sofh_in, float *sofh_out, float *cv, float *corr,
                                   const SyncTail &t, int n, int F, float alpha, int vec_width, hipStream_t s);
hipError_t sync_metric_launch(const float *cor_sof, const float *sofh_in, float *sofh_out, const float *cor_plsc, float *cv, float *corr,
                              const SyncTail &t, int n, int F, float alpha, int vec_width, hipStream_t s);
hipError_t sff_lr_launch(const float *X, float *Y, float *R_l, float *tmp, float *FRQ, float *PHS, int n, int F, float alpha, uint32_t *err_dev, hipStream_t s);
hipError_t sff_lr_recover(const float *X, float *Y, float *tmp, int n, int F, hipStream_t s);
hipError_t sff_fp_launch(const float *X, float *Y, float *tmp, float *FRQ, float *PHS, int n, int F, hipStream_t s);
hipError_t sync_vdelay_launch(const float *X, const float *Yprev, float *Yprev_new, float *Y, const float *buff_old, float *buff_new, const int *st_old, int *st_new,
                              const int32_t *Dtab, int n, int nbuff2, int F, hipStream_t s, int32_t *list = nullptr, const float **src = nullptr);

}  // namespace dvbs2
