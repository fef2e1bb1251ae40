/*
 * dvbs2hip.h -- C ABI of libdvbs2hip.so: the DVB-S2 RX inner path on AMD MI355X (gfx950).
 *
 * Drop-in boundary: each entry point replaces the body of ONE task codelet
 * of the reference (aff3ct/dvbs2).  The "replaces" lines cite the reference interface
 * (path:line relative to /root/reference).  Sockets of the reference hold n_frames
 * frames contiguously (frame f at offset f * n_elmts); so do all buffers here.
 *
 * Conventions
 *   - plain C, opaque handle, no exceptions across the ABI
 *   - return value: 0 = success, < 0 = dvbs2hip_status error (text: dvbs2hip_last_error)
 *   - B = int32_t (one bit per int, 0/1), R = Q = float, as in the reference (H6)
 *   - entry points WITHOUT suffix take HOST pointers (socket semantics): H2D copy,
 *     kernels, D2H copy, stream synchronised on return
 *   - entry points WITH `_dev` take DEVICE pointers, enqueue on the handle's stream and
 *     return without synchronising (chained tasks avoid PCIe round trips)
 *   - 1 <= n_frames <= cfg.max_frames; inter-frame batching (-F) maps to the grid width
 *   - a handle is entered by one host thread at a time (a StreamPU module instance is
 *     only ever entered by one thread: SURVEY.md 8b "Threading")
 *   - there is NO CPU fallback: without a usable HIP device dvbs2hip_create fails
 */
#ifndef DVBS2HIP_H
#define DVBS2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVBS2HIP_VERSION 101

typedef enum {
    DVBS2HIP_OK           = 0,
    DVBS2HIP_EINVAL       = -1,  /* spu::tools::invalid_argument / length_error        */
    DVBS2HIP_ENOMEM       = -2,  /* spu::tools::cannot_allocate                        */
    DVBS2HIP_EHIP         = -3,  /* spu::tools::runtime_error (HIP runtime failure)    */
    DVBS2HIP_EUNSUPPORTED = -4,  /* spu::tools::unimplemented_error                    */
    DVBS2HIP_ENODEVICE    = -5,  /* no HIP device: the product never falls back to CPU */
    DVBS2HIP_ETIMEOUT     = -6   /* dvbs2hip_monitor_reduce: a peer rank did not arrive within the timeout given at _reduce_init (it died): the communicator is
                                    aborted, the handle's stream is no longer usable -- the caller should leave with a non-zero exit code (a fresh exit, no re-exec) */
} dvbs2hip_status;

typedef struct dvbs2hip_handle dvbs2hip_t;

/* LDPC check-node rule (--dec-implem, DVBS2.cpp:117-149; the reference's default is SPA) */
enum { DVBS2HIP_IMPLEM_NMS = 0, DVBS2HIP_IMPLEM_MS = 1,
       DVBS2HIP_IMPLEM_SPA = 2,          /* sum-product as the reference decodes it: the exact check node with every message clipped at 2 atanh(1 - FLT_EPSILON) = 16.64, where
                                            the messages of AFF3CT's Update_rule_SPA saturate (fp32 tanh product) -- frame for frame the decisions of SPA_TANH on 99.9 % of the
                                            frames near the waterfall (results/r06/spa_rules.md); within 1e-4 max(1, |L|) of the oracle's ORC_SPA_CLIP */
       DVBS2HIP_IMPLEM_SPA_TANH = 3,   /* sum-product as AFF3CT's Update_rule_SPA evaluates it [UPSTREAM-RECALL]: tanh product in fp32, messages capped at
                                            2 atanh(1 - FLT_EPSILON) = 16.64; BIT-EXACT against the oracle's ORC_SPA_TANH (every operation correctly rounded) */
       DVBS2HIP_IMPLEM_SPA_EXACT = 4 };  /* the exact check node without the clip (rounds 1-5's SPA): within 1e-4 max(1, |L|) of the oracle's ORC_SPA; loses 20-50 % more frames than the
                                            reference at the low-FER end of the rate-3/5 traces and has an error floor below them (results/r06/spa_rules.md) */
/* Order in which the layered decoder visits the checks of a frame (dvbs2hip_set_ldpc_schedule) */
enum { DVBS2HIP_SCHED_QC = 0,        /* quasi-cyclic layers of 360 independent checks: the throughput path */
       DVBS2HIP_SCHED_NATURAL = 1 }; /* natural row order of H, what AFF3CT's BP_HORIZONTAL_LAYERED runs: one lane per frame */
/* Interleaver read order (DVBS2.cpp:300-317) */
enum { DVBS2HIP_ITL_TOP_LEFT = 0, DVBS2HIP_ITL_TOP_RIGHT = 1 };

/*
 * Configuration = the arguments the reference passes to its build_* functions
 * (DVBS2.cpp:287-356 modcod_init, :406-495 builders).  Pointers are only read during
 * dvbs2hip_create.  dvbs2hip_cfg_from_modcod fills everything for a named MODCOD.
 */
typedef struct dvbs2hip_cfg {
    /* code sizes */
    int32_t N_ldpc;            /* 16200 | 64800                                   */
    int32_t K_ldpc;            /* = N_bch                                         */
    int32_t K_bch;
    /* LDPC: ETSI EN 302 307 Annex B/C address table, K_ldpc/360 rows              */
    int32_t        ldpc_n_rows;
    const int32_t *ldpc_row_ptr;   /* n_rows + 1                                  */
    const int32_t *ldpc_addr;
    int32_t ldpc_n_ite;        /* --dec-ite                                       */
    int32_t ldpc_implem;       /* DVBS2HIP_IMPLEM_*                               */
    float   ldpc_alpha;        /* NMS normalisation factor; 1.0 = AFF3CT default  */
    int32_t ldpc_early_stop;   /* 1: stop on zero syndrome (enable_syndrome)      */
    /* BCH: GF(2^m) primitive polynomial (coefficient of x^i at [i], m+1 entries), t */
    int32_t        bch_m;
    int32_t        bch_t;
    const int32_t *bch_prim;
    /* modem: raw constellation (2^bps points, re/im interleaved, conf/mod order);  */
    /* normalised to unit mean energy inside, like tools::Constellation_user        */
    int32_t      bps;
    const float *cstl;
    /* bit interleaver: itl_cols <= 1 means Interleaver_core_NO                     */
    int32_t itl_cols;
    int32_t itl_order;
    /* matched filter: real taps (as Filter_RRC_ccr_naive::compute_rrc_coefs returns */
    /* them); fir_n_taps = 0 disables the filter entry points                       */
    int32_t      fir_n_taps;
    const float *fir_taps;
    int32_t      fir_osf;      /* samples per symbol of the filter input           */
    /* runtime */
    int32_t max_frames;        /* capacity of one call (the -F of the socket), 1 .. 65534.  A handle for at most one frame per CU (<= 256) decodes N = 64800 with the
                                  one-frame-per-CU kernel: lower latency per call (0.42 against 0.54 ms), same results bit for bit */
    int32_t device;            /* HIP device ordinal                               */
    void   *stream;            /* hipStream_t to enqueue on, or NULL: own stream   */
    int32_t ldpc_lds_groups;   /* tuning: < 0 = automatic                          */
    /* PLS code of the PLHEADER: the 7-entry mod_cod vector of Framer.hxx:111-126 (TX side only) */
    int32_t pls[7];
} dvbs2hip_cfg;

/* ------------------------------------------------------------------ lifecycle */
/* Names accepted: the reference's five ("QPSK-S_8/9", "QPSK-S_3/5", "8PSK-S_3/5",
 * "8PSK-S_8/9", "16APSK-S_8/9"; "" = "QPSK-S_8/9": DVBS2.cpp:287-319) plus the extension
 * rows "QPSK-N_8/9", "8PSK-N_8/9", "16APSK-N_8/9", "32APSK-S_3/4".  Unknown name:
 * DVBS2HIP_EINVAL (the reference throws invalid_argument, DVBS2.cpp:319).  Fills code
 * sizes, tables, constellation, interleaver and the 81-tap SRRC (Shaping_filter.hpp:24-28);
 * implem = SPA, n_ite = 50, alpha = 1, early_stop = 1, max_frames = 1 (the reference's defaults,
 * DVBS2.cpp:135-142). */
int dvbs2hip_cfg_from_modcod(const char *modcod, dvbs2hip_cfg *cfg);
int dvbs2hip_create(const dvbs2hip_cfg *cfg, dvbs2hip_t **out);
/* GPUs this process sees (0 without one): what a one-process-per-GPU launcher takes `cfg.device = local_rank % count` from when the
 * environment gives it a global rank only (host/dvbs2_tx_rx_bb.cpp); the reference's analogue is the thread count of Sequence(first, n_threads),
 * TX_RX_BB/main.cpp:96. */
int dvbs2hip_device_count(int32_t *count);
void dvbs2hip_destroy(dvbs2hip_t *h);
/* text of the last error on this handle (or of the last failed create when h == NULL) */
const char *dvbs2hip_last_error(const dvbs2hip_t *h);
/* selects the check order of ldpc_decode_siho / rx_bb on this handle (default DVBS2HIP_SCHED_QC).  NATURAL reproduces the
 * reference's horizontal-layered sweep (checks 0 .. M-1 in row order) exactly as the oracle's ORC_SCHED_NATURAL does; it
 * keeps 4 (N + 3 M) bytes of state per frame in HBM (22 MB per 64 normal frames, allocated on first use for max_frames)
 * and only pays off on batches of tens of thousands of frames.  NMS / MS only: DVBS2HIP_EUNSUPPORTED otherwise. */
int dvbs2hip_set_ldpc_schedule(dvbs2hip_t *h, int32_t schedule);
/* name of the LDPC kernel instantiation the plan selected for this MODCOD (diagnostics, bench.py's roofline line) */
const char *dvbs2hip_ldpc_kernel_name(const dvbs2hip_t *h);
/* Pins a host socket buffer (hipHostRegister) for the lifetime of the handle or until _unregister.  With every socket of a
 * call pinned, dvbs2hip_ldpc_decode_siho and dvbs2hip_rx_bb copy at PCIe speed and run the batch in chunks with the two copy
 * directions and the kernels overlapped (host-form QPSK normal frames: 29 k -> ~150 k frames/s); unpinned sockets keep the plain
 * copy -> kernels -> copy path.  StreamPU socket buffers live as long as their module, so an integration registers each once
 * after binding.  The buffer must stay allocated while registered. */
int dvbs2hip_host_register(dvbs2hip_t *h, void *ptr, size_t bytes);
int dvbs2hip_host_unregister(dvbs2hip_t *h, void *ptr);
/* Interface_reset: clears the filter state and the monitor counters */
int dvbs2hip_reset(dvbs2hip_t *h);
/* change --dec-ite / alpha / early-stop without rebuilding tables */
int dvbs2hip_set_ldpc_params(dvbs2hip_t *h, int32_t n_ite, float alpha, int32_t early_stop);
void *dvbs2hip_get_stream(dvbs2hip_t *h);      /* hipStream_t */
int dvbs2hip_synchronize(dvbs2hip_t *h);
/* A sequence of _dev calls as ONE submission (hipGraph; BASELINE configs[4], small-frame latency: at -F 1 the sequence filter -> extract -> rx_bb is six launches and a fill
 * around one workgroup's ten iterations).  Between _begin and _end the _dev calls of this handle are RECORDED, not run (same pointers, same n_frames on every replay; no host-
 * socket form, no dvbs2hip_synchronize in between; _begin returns DVBS2HIP_EINVAL while dvbs2hip_timing_enable is on; run the sequence once before recording it: first calls allocate).  Tasks that keep a memory from call to call (filter, shape_filter, the synchronizers) are recorded with the memory buffers of that moment: every replay starts from the state the stream had when the
 * sequence was recorded, which is what a latency measurement on one frame wants and NOT a way to run a stream; dvbs2hip_sync_coarse_synchronize_dev, whose stream position is a launch argument, refuses to be recorded.  The reference's counterpart is the task
 * sequence itself (RX/main_sched.cpp:199-223: a spu::runtime::Sequence runs its bound tasks back to back). */
int dvbs2hip_graph_begin(dvbs2hip_t *h);
int dvbs2hip_graph_end(dvbs2hip_t *h, int32_t *graph);
int dvbs2hip_graph_launch(dvbs2hip_t *h, int32_t graph);
int dvbs2hip_graph_destroy(dvbs2hip_t *h, int32_t graph);
/* derived sizes (per frame), as DVBS2.cpp:351-355 */
typedef struct dvbs2hip_sizes {
    int32_t N_ldpc, K_ldpc, K_bch, bps, N_xfec_sym, pl_frame_sym, ldpc_edges, ldpc_q;
} dvbs2hip_sizes;
int dvbs2hip_get_sizes(const dvbs2hip_t *h, dvbs2hip_sizes *out);

/* ------------------------------------------------------------------ a1  LDPC decoder
 * replaces: module::Decoder_SIHO::decode_siho(Y_N, CWD, V_K) of the decoder returned by
 * tools::Codec_LDPC<B,Q>::get_decoder_siho() -- built DVBS2.cpp:418-449, bound
 * TX_RX_BB/main.cpp:65,90-91 (sockets dec::sck::decode_siho::{Y_N,CWD,V_K}).
 *   Y_N : float  [n_frames * N_ldpc]  channel LLRs, LLR > 0 <=> bit 0
 *   CWD : int8_t [n_frames]           1 = codeword detected (zero syndrome)
 *   V_K : int32_t[n_frames * K_ldpc]  hard decisions of the systematic bits        */
int dvbs2hip_ldpc_decode_siho(dvbs2hip_t *h, const float *Y_N, int8_t *CWD, int32_t *V_K, int32_t n_frames);
int dvbs2hip_ldpc_decode_siho_dev(dvbs2hip_t *h, const float *Y_N, int8_t *CWD, int32_t *V_K, int32_t n_frames);
/* test hook: also returns the N_ldpc posteriors (natural order) and the iteration count */
int dvbs2hip_ldpc_decode_siho_post(dvbs2hip_t *h, const float *Y_N, int8_t *CWD, int32_t *V_K,
                                   float *post_N, int32_t *n_ite_done, int32_t n_frames);

/* ------------------------------------------------------------------ a2  BCH decoder
 * replaces: Decoder_BCH_DVBS2<B,R>::_decode_hiho(const B *Y_N, int8_t *CWD, B *V_K, frame_id)
 * -- src/common/Module/Decoder_BCH_DVBS2/Decoder_BCH_DVBS2.cpp:28-40 (bit reversal around
 * Decoder_BCH_std::_decode), built DVBS2.cpp:406-416, bound TX_RX_BB/main.cpp:91-92.
 *   Y_N : int32_t[n_frames * K_ldpc]   V_K : int32_t[n_frames * K_bch]   CWD = !status */
int dvbs2hip_bch_decode_hiho(dvbs2hip_t *h, const int32_t *Y_N, int8_t *CWD, int32_t *V_K, int32_t n_frames);
int dvbs2hip_bch_decode_hiho_dev(dvbs2hip_t *h, const int32_t *Y_N, int8_t *CWD, int32_t *V_K, int32_t n_frames);

/* ------------------------------------------------------------------ a3  soft demapper
 * replaces: module::Modem_generic_fast<B,R,Q,max_star>::demodulate(CP, Y_N1, Y_N2)
 * -- built DVBS2.cpp:478-488, bound TX_RX_BB/main.cpp:87-88.
 *   CP  : float[n_frames]              sigma (per real dimension), one per frame
 *   Y_N1: float[n_frames * 2*N_xfec]   symbols, re/im interleaved
 *   Y_N2: float[n_frames * N_ldpc]     LLRs in transmission (interleaved) order      */
int dvbs2hip_demodulate(dvbs2hip_t *h, const float *CP, const float *Y_N1, float *Y_N2, int32_t n_frames);
int dvbs2hip_demodulate_dev(dvbs2hip_t *h, const float *CP, const float *Y_N1, float *Y_N2, int32_t n_frames);

/* ------------------------------------------------------------------ a4  LLR de-interleaver
 * replaces: module::Interleaver<float,uint32_t>::deinterleave(itl, nat)
 * -- DVBS2.cpp:451-476, bound TX_RX_BB/main.cpp:56,89.                              */
int dvbs2hip_deinterleave(dvbs2hip_t *h, const float *itl, float *nat, int32_t n_frames);
int dvbs2hip_deinterleave_dev(dvbs2hip_t *h, const float *itl, float *nat, int32_t n_frames);
/* a3+a4 fused (one pass, permuted store) */
int dvbs2hip_demodulate_deinterleave(dvbs2hip_t *h, const float *CP, const float *Y_N1, float *nat, int32_t n_frames);
int dvbs2hip_demodulate_deinterleave_dev(dvbs2hip_t *h, const float *CP, const float *Y_N1, float *nat, int32_t n_frames);

/* ------------------------------------------------------------------ a5  SRRC matched filter
 * replaces: Filter<R>::filter(X_N1, Y_N2) -> Filter_FIR_ccr<R>::_filter
 * -- src/common/Module/Filter/Filter_FIR/Filter_FIR_ccr.cpp:68-142 (+ step(),
 * Filter_FIR_ccr.hpp:39-52), bound RX/main_sched.cpp:199-201.  The n_frames frames of
 * one call are consecutive in time; the last (n_taps-1) complex samples are kept in the
 * handle between calls (Filter_FIR_ccr.cpp:80-83); dvbs2hip_filter_reset zeroes them.
 *   X_N1, Y_N2 : float[n_frames * 2 * n_cplx]                                        */
int dvbs2hip_filter(dvbs2hip_t *h, const float *X_N1, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_filter_dev(dvbs2hip_t *h, const float *X_N1, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_filter_reset(dvbs2hip_t *h);
/* replaces: Filter<R>::filter1(X_N1, Y_N2) and ::filter2(X_N1, Y_N2h, Y_N2) -> Filter_FIR_ccr<R>::_filter1 / _filter2
 * -- Filter_FIR_ccr.cpp:144-218 and :220-294; tasks and sockets flt::tsk::{filter1,filter2}, flt::sck::filter1::{X_N1,Y_N2},
 * flt::sck::filter2::{X_N1,Y_N2h,Y_N2} (Filter.hpp:22-29, codelets Filter.hxx:69-96); bound RX/main_sched.cpp:199-201 and split
 * over two pipeline stages in main_13-sta.cpp:274-282.  The reference's filter1 writes the outputs below about half a frame and
 * advances the filter state; its filter2 copies Y_N2h and computes the rest from X_N1 alone.  Same here, as pure functions of
 * the sockets (the two tasks may run in different pipeline stages on different batches):
 *   filter1: Y_N2[f][0 .. split) = the filtered frame, state advanced exactly as dvbs2hip_filter does.  (The samples from
 *            `split` on, which the reference leaves as they were, are filled as well -- nothing may depend on them.)
 *   filter2: Y_N2[f][0 .. split) = Y_N2h[f][0 .. split);  Y_N2[f][split .. n_cplx) computed from X_N1[f] only; no state read
 *            or written.  filter2(X, filter1(X)) == filter(X) bit for bit.
 * split = dvbs2hip_filter_split(h, n_cplx) complex samples (n_cplx / 2 rounded down to a multiple of 4; the reference's own
 * split depends on its SIMD width).  Half a frame has to hold the filter's memory: n_cplx / 2 >= n_taps - 1, else EINVAL
 * (the reference reads out of bounds in that case).   X_N1, Y_N2h, Y_N2 : float[n_frames * 2 * n_cplx]                  */
int dvbs2hip_filter_split(const dvbs2hip_t *h, int32_t n_cplx);
int dvbs2hip_filter1(dvbs2hip_t *h, const float *X_N1, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_filter1_dev(dvbs2hip_t *h, const float *X_N1, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_filter2(dvbs2hip_t *h, const float *X_N1, const float *Y_N2h, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_filter2_dev(dvbs2hip_t *h, const float *X_N1, const float *Y_N2h, float *Y_N2, int32_t n_cplx, int32_t n_frames);
/* Kernel behind dvbs2hip_filter[_dev] on this handle.  AUTO (default): the matrix-core form (bf16 x 3 split operands,
 * fp32 accumulation, k_fir_mfma.hip) for filters of at most 81 taps on 16-byte aligned sockets, the fp32 vector kernel
 * otherwise; VALU forces the vector kernel; MFMA returns DVBS2HIP_EUNSUPPORTED for a longer filter.  Both meet the same
 * 1e-4 bar against Filter_FIR_ccr's fp32 FMA chain (neither reproduces its rounding order bit for bit). */
enum { DVBS2HIP_FIR_AUTO = 0, DVBS2HIP_FIR_VALU = 1, DVBS2HIP_FIR_MFMA = 2 };
int dvbs2hip_set_filter_kernel(dvbs2hip_t *h, int32_t kernel);

/* ------------------------------------------------------------------ a6  noise estimator
 * replaces: Estimator<R>::estimate(X_N, SIG, Eb_N0, Es_N0) -> Estimator_DVBS2<R>::_estimate
 * -- src/common/Module/Estimator/Estimator_DVBS2.hxx:31-58, wrapper Estimator.hxx:103-118.
 *   X_N : float[n_frames * 2*N_xfec]; SIG, Eb_N0, Es_N0 : float[n_frames]            */
int dvbs2hip_estimate(dvbs2hip_t *h, const float *X_N, float *SIG, float *Eb_N0, float *Es_N0, int32_t n_frames);
int dvbs2hip_estimate_dev(dvbs2hip_t *h, const float *X_N, float *SIG, float *Eb_N0, float *Es_N0, int32_t n_frames);

/* ------------------------------------------------------------------ the gain stages of the RX graph
 * replaces: Multiplier_AGC_cc_naive::imultiply -> _imultiply
 * -- src/common/Module/Multiplier/Sequence/Multiplier_AGC_cc_naive.cpp:22-46 (task / sockets Multiplier.hxx:49-60): every frame is divided by its standard deviation
 * about its mean over sqrt(output_energy).  The reference builds two: `front_agc` on the received samples (n_cplx = pl_frame * osf, output_energy = 1 / osf: DVBS2.cpp:660-664,
 * bound RX/main_sched.cpp:197-198) and `mult_agc` on the symbols behind the timing synchronizer, in front of the frame synchronizer (n_cplx = pl_frame, output_energy = 1:
 * DVBS2.cpp:653-657, bound RX/main_sched.cpp:205-207).  A frame of equal values divides by zero as it does there.
 *   X_N: float[n_frames * 2*n_cplx]  ->  Z_N: same size                                                   */
int dvbs2hip_agc_imultiply(dvbs2hip_t *h, const float *X_N, float *Z_N, int32_t n_cplx, float output_energy, int32_t n_frames);
int dvbs2hip_agc_imultiply_dev(dvbs2hip_t *h, const float *X_N, float *Z_N, int32_t n_cplx, float output_energy, int32_t n_frames);

/* The coarse frequency synchronizer's task in the TRANSMISSION phase: the frequency shift alone
 * replaces: Synchronizer_freq_coarse::synchronize -> Synchronizer_freq_coarse_DVBS2_aib::_synchronize = Multiplier_sine_ccc_naive::imultiply
 * -- src/common/Module/Synchronizer/Synchronizer_freq/Synchronizer_freq_coarse/Synchronizer_freq_coarse_DVBS2_aib.cpp:43-50, Multiplier/Sine/Multiplier_sine_ccc_naive.cpp:69-77
 * (sockets X_N1 / FRQ / PHS / Y_N2: Synchronizer_freq_coarse.hxx:35-39; bound RX/main_sched.cpp:198-200).  z_n = x_n exp(j omega n), n the position in the stream (it starts
 * over after 999999; the frequency is kept to six decimals so that this is seamless).  The LOOP that finds the frequency (update_phase, :53-95, one step per pilot symbol fed back
 * from the timing synchronizer) is sample-serial and out of scope: _set_freq takes what it, or anything else, has estimated.  FRQ / PHS per frame: that frequency and 0.
 *   X_N1: float[n_frames * 2*n_cplx] -> Y_N2: same size (frames = consecutive stretches of ONE stream)             */
int dvbs2hip_sync_coarse_set_freq(dvbs2hip_t *h, float estimated_freq);
int dvbs2hip_sync_coarse_reset(dvbs2hip_t *h);
int dvbs2hip_sync_coarse_synchronize(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_sync_coarse_synchronize_dev(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t n_cplx, int32_t n_frames);

/* ------------------------------------------------------------------ a7  PL descramble, header/pilot removal
 * replaces: Scrambler_PL<D>::descramble -> __scramble(scr_flag = false)
 * -- src/common/Module/Scrambler/Scrambler_PL/Scrambler_PL.hxx:61-78 (start_ix = 90)
 *   Y_N1, Y_N2 : float[n_frames * 2*pl_frame]                                        */
int dvbs2hip_pl_descramble(dvbs2hip_t *h, const float *Y_N1, float *Y_N2, int32_t n_frames);
int dvbs2hip_pl_descramble_dev(dvbs2hip_t *h, const float *Y_N1, float *Y_N2, int32_t n_frames);
/* replaces: Framer<B>::remove_plh -> _remove_plh -- src/common/Module/Framer/Framer.hxx:330-343
 *   Y_N1 : float[n_frames * 2*pl_frame] -> Y_N2 : float[n_frames * 2*N_xfec]         */
int dvbs2hip_remove_plh(dvbs2hip_t *h, const float *Y_N1, float *Y_N2, int32_t n_frames);
int dvbs2hip_remove_plh_dev(dvbs2hip_t *h, const float *Y_N1, float *Y_N2, int32_t n_frames);

/* ------------------------------------------------------------------ a8  BB descramble
 * replaces: Scrambler_BB<D>::_descramble -- src/common/Module/Scrambler/Scrambler_BB/Scrambler_BB.hxx:51-72
 *   Y_N1, Y_N2 : int32_t[n_frames * K_bch]                                           */
int dvbs2hip_bb_descramble(dvbs2hip_t *h, const int32_t *Y_N1, int32_t *Y_N2, int32_t n_frames);
int dvbs2hip_bb_descramble_dev(dvbs2hip_t *h, const int32_t *Y_N1, int32_t *Y_N2, int32_t n_frames);

/* ------------------------------------------------------------------ a9  BER/FER monitor
 * replaces: module::Monitor_BFER<B>::check_errors(U, V) -- built DVBS2.cpp:575-591, bound
 * TX_RX_BB/main.cpp:93-94.  Counters {FRA, BE, FE} live on the device; get copies them out
 * (what tools::Monitor_reduction sums across threads, main.cpp:123-125, is summed across
 * GPUs by the host with one 24-byte all-reduce: DESIGN.md "multi-GPU").
 *   U, V : int32_t[n_frames * K_bch]                                                 */
int dvbs2hip_monitor_check_errors(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int32_t n_frames);
int dvbs2hip_monitor_check_errors_dev(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int32_t n_frames);
int dvbs2hip_monitor_get(dvbs2hip_t *h, uint64_t fra_be_fe[3]);
int dvbs2hip_monitor_reset(dvbs2hip_t *h);
/* replaces: module::Monitor_BFER<B>::check_errors2(U, V, FRA, BE, FE, BER, FER) -- the task the RX mains bind
 * (RX/main_sched.cpp:222-223 U / V, :244-247 BE / FE / BER / FER into probes; aff3ct, absent: sockets
 * mnt::sck::check_errors2::{U,V,FRA,BE,FE,BER,FER}).  Counts like check_errors AND writes the monitor's counters as they
 * stand after each frame of the call (frame f of the socket = state after frames 0..f): FRA int64, BE / FE int32,
 * BER = BE / FRA / K_bch and FER = FE / FRA as float, with Monitor_BFER's convention that before the first bit error they
 * report the bound 1 / FRA [/ K_bch] instead of 0.  Any of the five output pointers may be NULL.
 *   U, V : int32_t[n_frames * K_bch];  FRA : int64_t[n_frames];  BE, FE : int32_t[n_frames];  BER, FER : float[n_frames]   */
int dvbs2hip_monitor_check_errors2(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int64_t *FRA, int32_t *BE, int32_t *FE,
                                   float *BER, float *FER, int32_t n_frames);
int dvbs2hip_monitor_check_errors2_dev(dvbs2hip_t *h, const int32_t *U, const int32_t *V, int64_t *FRA, int32_t *BE, int32_t *FE,
                                       float *BER, float *FER, int32_t n_frames);
/* replaces: tools::Monitor_reduction over the per-thread monitors (TX_RX_BB/main.cpp:123-125 construction + 500 ms period,
 * :155-161 is_done_all / final reduction) for ONE PROCESS PER GPU: the sum of {FRA, BE, FE} over the ranks, one RCCL
 * all-reduce of 3 x uint64 on the handle's stream (over xGMI inside a node).  COLLECTIVE: every rank calls _reduce the same
 * number of times (call it once per batch, as is_done_all is).  librccl.so is opened at run time on the first _init.
 *   init    : rank 0 creates the communicator id and hands it to the other ranks through files named after `rendezvous_path`
 *             (dvbs2hip_rendezvous below); each side waits for at most timeout_ms (< 0: for ever).  world_size 1 needs no file.
 *             ncclCommInitRank itself has no timeout: a launcher has to end the whole job when one rank dies.
 *   reduce  : the reduced counters; on a handle without _init it is dvbs2hip_monitor_get (a single process).  With timeout_ms > 0 at _init a reduction whose peers do
 *             not arrive within timeout_ms returns DVBS2HIP_ETIMEOUT (the communicator is aborted; leave with a non-zero exit code) instead of waiting for ever.
 *   finalize: destroys the communicator (also done by dvbs2hip_destroy).                                              */
int dvbs2hip_monitor_reduce_init(dvbs2hip_t *h, int32_t rank, int32_t world_size, const char *rendezvous_path, int32_t timeout_ms);
/* the out-of-band step of _reduce_init on its own (no GPU needed): rank 0 hands `bytes` bytes of `blob` to every other rank, which
 * receives them in `blob`.  Files left by an earlier run are harmless: rank r publishes a fresh random nonce in `<path>.hello.<r>` and takes
 * the payload only from a `<path>.ack.<r>` that repeats it; a completed exchange leaves no file behind.  Give every job a path of its own
 * in a directory only its user can write.  0, DVBS2HIP_EINVAL (bad arguments, cannot write) or DVBS2HIP_EHIP (timed out).                */
int dvbs2hip_rendezvous(int32_t rank, int32_t world_size, const char *rendezvous_path, void *blob, size_t bytes, int32_t timeout_ms);
int dvbs2hip_monitor_reduce(dvbs2hip_t *h, uint64_t fra_be_fe[3]);
int dvbs2hip_monitor_reduce_finalize(dvbs2hip_t *h);

/* ------------------------------------------------------------------ fused RX baseband chain (a7 -> a8)
 * One call = the RX half of TX_RX_BB/main.cpp:83-92:
 *   PL descramble -> remove_plh -> estimate -> demodulate -> deinterleave -> LDPC
 *   decode_siho -> BCH decode_hiho -> BB descramble, all intermediates device-resident.
 *   pl_frames: float  [n_frames * 2*pl_frame]
 *   sigma_in : float  [n_frames] or NULL (NULL: Estimator_DVBS2; else Estimator_perfect)
 *   info_bits: int32_t[n_frames * K_bch]
 *   cwd_ldpc, cwd_bch: int8_t[n_frames] (either may be NULL)                         */
int dvbs2hip_rx_bb(dvbs2hip_t *h, const float *pl_frames, const float *sigma_in, int32_t *info_bits,
                   int8_t *cwd_ldpc, int8_t *cwd_bch, int32_t n_frames);
int dvbs2hip_rx_bb_dev(dvbs2hip_t *h, const float *pl_frames, const float *sigma_in, int32_t *info_bits,
                       int8_t *cwd_ldpc, int8_t *cwd_bch, int32_t n_frames);
/* the fused chain reading frame f where SRC[f] points (the table dvbs2hip_sync_frame_locate_dev fills): same outputs as dvbs2hip_rx_bb_dev on the delayed copy */
int dvbs2hip_rx_bb_located_dev(dvbs2hip_t *h, const float *const *SRC, const float *sigma, int32_t *info_bits, int8_t *cwd_ldpc, int8_t *cwd_bch, int32_t n_frames);

/* ------------------------------------------------------------------ N1: TX mirror + AWGN channel (test-signal side)
 * One call = the TX half + channel of TX_RX_BB/main.cpp:75-82 on the device:
 *   Source::generate -> Scrambler_BB::scramble -> Encoder_BCH_DVBS2::encode
 *   (src/common/Module/Encoder_BCH_DVBS2/Encoder_BCH_DVBS2.cpp:28-43) -> LDPC encode (DVBS2.cpp:427)
 *   -> Interleaver::interleave -> Modem::modulate -> Framer::generate (Framer.hxx:232-293)
 *   -> Scrambler_PL::scramble (Scrambler_PL.hxx:61-78) -> Channel_AWGN::add_noise (DVBS2.cpp:593-613).
 *   info_in  : int32_t[n_frames * K_bch] or NULL (NULL: Source_random, Philox keyed by seed)
 *   sigma    : float[n_frames] noise std-dev per real dimension, or NULL for no noise
 *   info_out : int32_t[n_frames * K_bch] the payload that was sent (the monitor's U socket), or NULL
 *   pl_frames: float[n_frames * 2*pl_frame]                                             */
int dvbs2hip_tx_bb(dvbs2hip_t *h, const int32_t *info_in, uint64_t seed, const float *sigma, int32_t *info_out,
                   float *pl_frames, int32_t n_frames);
int dvbs2hip_tx_bb_dev(dvbs2hip_t *h, const int32_t *info_in, uint64_t seed, const float *sigma, int32_t *info_out,
                       float *pl_frames, int32_t n_frames);

/* ------------------------------------------------------------------ N2: TX shaping filter, channel noise, perfect timing
 * replaces: Filter_UPRRC_ccr_naive::filter -> Filter_UPFIR_ccr_naive::_filter
 * -- src/common/Module/Filter/Filter_UPFIR/Filter_UPFIR_ccr_naive.cpp:52-66 (polyphase bank of `fir_osf` FIRs built
 * from the handle's taps), bound src/mains/TX_RX/main.cpp:212,223.  Keeps (n_taps-1)/osf input samples between calls.
 *   X_N1: float[n_frames * 2*n_cplx]  ->  Y_N2: float[n_frames * 2*n_cplx*osf]                    */
int dvbs2hip_shape_filter(dvbs2hip_t *h, const float *X_N1, float *Y_N2, int32_t n_cplx, int32_t n_frames);
int dvbs2hip_shape_filter_dev(dvbs2hip_t *h, const float *X_N1, float *Y_N2, int32_t n_cplx, int32_t n_frames);
/* replaces: Channel_AWGN::add_noise(CP, X_N, Y_N) -- built DVBS2.cpp:593-613, bound TX_RX_BB/main.cpp:81-82.
 *   CP: float[n_frames] sigma per real value; X_N, Y_N: float[n_frames * n_elmts] (n_elmts even)   */
int dvbs2hip_add_noise(dvbs2hip_t *h, const float *CP, const float *X_N, float *Y_N, uint64_t seed, int32_t n_elmts, int32_t n_frames);
int dvbs2hip_add_noise_dev(dvbs2hip_t *h, const float *CP, const float *X_N, float *Y_N, uint64_t seed, int32_t n_elmts, int32_t n_frames);
/* perfect-timing extraction (Synchronizer_timing_perfect, DVBS2.cpp:558-570): the batch is ONE stream,
 * Y[i] = X[offset + i*osf] for i < n_frames*n_cplx_out; samples outside the batch read as zero.      */
int dvbs2hip_extract_dev(dvbs2hip_t *h, const float *X, float *Y, int32_t n_cplx_out, int32_t osf, int64_t offset, int32_t n_frames);

/* ------------------------------------------------------------------ N4: frame synchronizer
 * replaces: Synchronizer_frame_DVBS2_fast<R> (type "FAST", the factory default,
 * src/common/Factory/Module/Synchronizer_frame/Synchronizer_frame.cpp:71-72, .hpp:26-30) --
 * src/common/Module/Synchronizer/Synchronizer_frame/Synchronizer_frame_DVBS2_fast.cpp:
 *   synchronize1 (:132-150)  differential signal, correlators corr_SOF (25 taps) / corr_PLSC (64 taps)
 *   synchronize2 (:222-299)  cor_SOF delayed by 64, correlation metric, alpha-average, arg max -> delay,
 *                            variable output delay (Filter/Variable_delay/Variable_delay_cc_naive.cpp:56-79)
 *   synchronize  (:46-128)   both in one task
 * A call carries n_frames PL frames that are consecutive in time and behaves as n_frames calls of the
 * reference task with its n_frames = 1 (the reference's delay line re-reads its own output socket, so
 * its multi-frame form depends on what the socket held before; that case is not reproduced).  State
 * (correlator memories, reg_channel, corr_vec, delay lines, previous output frame) lives in the handle.
 *   X_N1: float[n_frames * 2*pl_frame]; cor_SOF, cor_PLSC, Y_N2: same size; DEL (delay), FLG (packet flag, may be
 *   NULL): int32_t[n_frames]; TRI (metric, may be NULL): float[n_frames] -- the sockets of Synchronizer_frame.hxx:40-81
 * set_params: alpha / trigger (defaults 0.9 / 30) and vec_width = mipp::N<float>() of the reference build
 * (default 8 = AVX2): the samples past the last full vector of a frame are not alpha-averaged (:284-285).
 * get_metric: _get_metric / _get_packet_flag (.hpp:59-60) after the last frame processed.            */
int dvbs2hip_sync_frame_set_params(dvbs2hip_t *h, float alpha, float trigger, int32_t vec_width);
int dvbs2hip_sync_frame_reset(dvbs2hip_t *h);
int dvbs2hip_sync_frame_synchronize1(dvbs2hip_t *h, const float *X_N1, float *cor_SOF, float *cor_PLSC, int32_t n_frames);
int dvbs2hip_sync_frame_synchronize1_dev(dvbs2hip_t *h, const float *X_N1, float *cor_SOF, float *cor_PLSC, int32_t n_frames);
int dvbs2hip_sync_frame_synchronize2(dvbs2hip_t *h, const float *X_N1, const float *cor_SOF, const float *cor_PLSC, int32_t *DEL, int32_t *FLG,
                                     float *TRI, float *Y_N2, int32_t n_frames);
int dvbs2hip_sync_frame_synchronize2_dev(dvbs2hip_t *h, const float *X_N1, const float *cor_SOF, const float *cor_PLSC, int32_t *DEL, int32_t *FLG,
                                         float *TRI, float *Y_N2, int32_t n_frames);
int dvbs2hip_sync_frame_synchronize(dvbs2hip_t *h, const float *X_N1, int32_t *DEL, int32_t *FLG, float *TRI, float *Y_N2, int32_t n_frames);
int dvbs2hip_sync_frame_synchronize_dev(dvbs2hip_t *h, const float *X_N1, int32_t *DEL, int32_t *FLG, float *TRI, float *Y_N2, int32_t n_frames);
/* (round 5) CHAINED device form for a consumer of this library (Synchronizer_frame_DVBS2_fast.cpp:296-299 materializes Y_N2 only because the next task is a separate
 * module): the same synchronizer -- same DEL / FLG / TRI, same state for the next call -- but instead of the delayed copy it returns where each aligned frame STARTS:
 * SRC[f] (device table of n_frames device pointers, 8-byte aligned frames of 2 * pl_frame floats) points into X_N1 when the frame is one run of the input stream (the
 * delay did not move from frame f - 1: every frame in lock but the first and the last of a call) and into scratch of the handle otherwise.  X_N1 and the table must stay
 * untouched until their consumer -- dvbs2hip_rx_bb_located_dev, on the same handle and stream -- has run; the next synchronizer call reuses the scratch.  Bit-exact by
 * construction: the delay line only copies.                                                    */
int dvbs2hip_sync_frame_locate_dev(dvbs2hip_t *h, const float *X_N1, int32_t *DEL, int32_t *FLG, float *TRI, const float **SRC, int32_t n_frames);
int dvbs2hip_sync_frame_get_metric(dvbs2hip_t *h, float *max_corr, int32_t *packet_flag);

/* ------------------------------------------------------------------ N4: fine frequency / phase synchronizers
 * replace: Synchronizer_freq_fine::synchronize (sockets X_N1, FRQ, PHS, Y_N2: Synchronizer_freq_fine.hxx:32-47) of
 *   Synchronizer_Luise_Reggiannini_DVBS2_aib -- src/common/Module/Synchronizer/Synchronizer_freq/Synchronizer_freq_fine/
 *     Synchronizer_Luise_Reggiannini_DVBS2_aib.cpp:93-167 (autocorrelation of the pilot blocks, damped by lr_alpha from
 *     frame to frame -- default 0.999, Factory/Module/Synchronizer_freq_fine/Synchronizer_freq_fine.hpp:25 -- then the
 *     frame is rotated by the estimated frequency); _reset :170-176
 *   Synchronizer_freq_phase_DVBS2_aib -- .../Synchronizer_freq_phase_DVBS2_aib.cpp:44-112 (phase of every pilot block,
 *     unwrapped, least-squares line -> frequency and phase, rotation of the frame); stateless
 * Input = PL-DESCRAMBLED frames (the RX mains run them after Scrambler_PL::descramble).
 *   X_N1, Y_N2: float[n_frames * 2*pl_frame]; FRQ, PHS: float[n_frames] (may be NULL)                         */
int dvbs2hip_sync_lr_synchronize(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t n_frames);
int dvbs2hip_sync_lr_synchronize_dev(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t n_frames);
int dvbs2hip_sync_lr_set_alpha(dvbs2hip_t *h, float alpha);
int dvbs2hip_sync_lr_reset(dvbs2hip_t *h);
/* L&R runs its serial recurrence and the rotation in ONE launch: the rotating workgroups wait for workgroup 0's estimates, which assumes that the
 * dispatcher places workgroup 0 first ("L&R dispatch order").  A workgroup that waits longer than one second of wall-clock time
 * drops its stores and sets an error word; the host-socket form repeats the rotation before it returns, the _dev form when dvbs2hip_synchronize
 * is called (it returns DVBS2HIP_EHIP only if the repeat itself fails).  This counts the launches that needed the repeat (0 on every box so far). */
int dvbs2hip_sync_lr_timeouts(dvbs2hip_t *h, int32_t *n_launches_repeated);
int dvbs2hip_sync_freq_phase_synchronize(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t n_frames);
int dvbs2hip_sync_freq_phase_synchronize_dev(dvbs2hip_t *h, const float *X_N1, float *FRQ, float *PHS, float *Y_N2, int32_t n_frames);

/* ------------------------------------------------------------------ measurement
 * Per-kernel device time, measured with hipEvents recorded on the handle's stream around
 * each launch while timing is enabled (the equivalent of `--sim-stats`,
 * TX_RX_BB/main.cpp:110,170-178).                                                    */
enum { DVBS2HIP_K_LDPC = 0, DVBS2HIP_K_BCH = 1, DVBS2HIP_K_DEMAP = 2, DVBS2HIP_K_FIR = 3,
       DVBS2HIP_K_FRONT = 4, DVBS2HIP_K_MISC = 5, DVBS2HIP_K_COUNT = 6 };
int dvbs2hip_timing_enable(dvbs2hip_t *h, int32_t on);
int dvbs2hip_timing_reset(dvbs2hip_t *h);
/* synchronises, then returns the summed device ms and the number of launches of kernel k */
int dvbs2hip_timing_get(dvbs2hip_t *h, int32_t k, double *total_ms, int64_t *n_launches);
/* what a plain device-to-device copy reaches on this GPU with a kernel of this library (16 bytes per lane and access; the best of 1 / 4 accesses
 * in flight per lane, plain / non-temporal): read + written bytes over the hipEvent time of `reps` copies of `bytes` bytes in scratch memory, in GB/s.  A measurement aid for the roofline figures (bench.py: `roofline.hbm_copy_GBps_measured`); the
 * reference has no counterpart.                                                        */
int dvbs2hip_device_copy_bandwidth(dvbs2hip_t *h, size_t bytes, int32_t reps, double *GBps);
/* device memory helpers so non-HIP hosts (ctypes, cgo, JNI) can own device buffers */
int dvbs2hip_malloc(dvbs2hip_t *h, void **dptr, size_t bytes);
int dvbs2hip_free(dvbs2hip_t *h, void *dptr);
int dvbs2hip_memcpy_h2d(dvbs2hip_t *h, void *dst, const void *src, size_t bytes);
int dvbs2hip_memcpy_d2h(dvbs2hip_t *h, void *dst, const void *src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* DVBS2HIP_H */
