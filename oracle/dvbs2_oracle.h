/*
 * dvbs2_oracle.h -- CPU ORACLE for the DVB-S2 RX inner path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libdvbs2hip.so) never links, loads or calls anything in here.
 *
 * PARITY STATUS: "parity unpinned" for a1 (LDPC), a2 (BCH core), a3 (demapper),
 * a4 (interleaver): the reference keeps that arithmetic in the git submodules
 * lib/aff3ct + lib/streampu, which are EMPTY in /root/reference, and it
 * ships no unit tests or golden vectors (F3).  Those parts restate the published
 * algorithms (ETSI EN 302 307 + the AFF3CT v3.0.2-era behaviour recalled in SURVEY.md
 * 3c) and are anchored on the reference's own call sites.  PINNED parts: the PL
 * scrambling sequence (bit-exact against the 66420-entry PL_RAND_SEQ table), the
 * PLHEADER constants, the frame sizes, the Es/N0<->Eb/N0 mapping of refs/, and --
 * statistically -- the FER rows of refs/TX_RX_BB (tests/golden/refs_tx_rx_bb.json).
 *
 * All path:line citations are relative to /root/reference.
 */
#ifndef DVBS2_ORACLE_H
#define DVBS2_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- LDPC (a1) */
typedef struct orc_ldpc orc_ldpc;
/* rows: K/360 address rows of the ETSI table, row i = addr[row_ptr[i]..row_ptr[i+1]) */
orc_ldpc *orc_ldpc_create(int N, int K, int n_rows, const int *row_ptr, const int *addr);
void      orc_ldpc_destroy(orc_ldpc *c);
int       orc_ldpc_n_edges(const orc_ldpc *c);
int       orc_ldpc_q(const orc_ldpc *c);
/* CSR of H by check, natural order (for tests): returns pointers owned by c */
void      orc_ldpc_csr(const orc_ldpc *c, const int **chk_ptr, const int **chk_var);
/* IRA encoder, ETSI EN 302 307 5.3.2 (enc type "LDPC_DVBS2", DVBS2.cpp:427) */
void      orc_ldpc_encode(const orc_ldpc *c, const int32_t *info, int32_t *cw);
/* syndrome weight of a hard word */
int       orc_ldpc_syndrome_weight(const orc_ldpc *c, const int32_t *cw);

/* ORC_SPA: exact boxplus (no saturation).  ORC_SPA_TANH: the tanh-product form with fp32 saturation as AFF3CT's Update_rule_SPA evaluates it
 * [UPSTREAM-RECALL], written with correctly rounded IEEE operations only so that the HIP kernels match it bit for bit (dvbs2_oracle.c). */
enum { ORC_NMS = 0, ORC_SPA = 1, ORC_SPA_TANH = 2, ORC_SPA_CLIP = 3 };      /* ORC_SPA_CLIP: ORC_SPA with |c->v| clipped to 2 atanh(1 - FLT_EPSILON) = 16.6355 */
enum { ORC_SCHED_NATURAL = 0, ORC_SCHED_QC = 1, ORC_SCHED_QC_SEQ = 2, ORC_SCHED_QC_FIX = 3 };      /* QC_SEQ: the QC layers' order of checks, processed one after the other (analysis aid) */
/*
 * Horizontal-layered BP (dec type "BP_HORIZONTAL_LAYERED", DVBS2.cpp:428), implem
 * NMS (alpha; alpha = 1 gives plain MS), SPA (exact) or SPA_TANH (AFF3CT's saturating form).
 *   sched NATURAL: checks swept in natural row order of H  (AFF3CT order)
 *   sched QC     : q layers of 360 independent checks (c mod q == layer), the schedule
 *                  the HIP kernel runs; same-layer double edges resolved by ordered
 *                  delta updates (DESIGN.md "QC-layer schedule").
 * llr > 0 <=> bit 0.  early_stop: stop after an iteration whose hard decision has a
 * zero syndrome.  post (may be NULL) receives the N posteriors in natural order.
 * returns the number of iterations run; *cwd = 1 iff final syndrome is zero.
 */
int orc_ldpc_decode(const orc_ldpc *c, const float *llr, int implem, int sched, int n_ite,
                    float alpha, int early_stop, int32_t *bits_K, float *post, int8_t *cwd);

/* test hooks */
float orc_det_tanh_half(float a);
float orc_det_log1p(float w);
void  orc_chk_update(int implem, float alpha, const float *v2c, int d, float *out);

/* ---------------------------------------------------------------- BCH (a2) */
typedef struct orc_bch orc_bch;
/* GF(2^m) from primitive polynomial bits prim[0..m] (prim[i] = coeff of x^i),
 * cf. tools::BCH_polynomial_generator<>(16383, 12, {1,1,0,1,0,1,0,0,0,0,0,0,0,0,1})
 * (TX_RX_BB/main.cpp:45, DVBS2.hpp:55); shortened (N,K) code, t errors. */
orc_bch *orc_bch_create(int m, const int *prim, int t, int N, int K);
void     orc_bch_destroy(orc_bch *b);
int      orc_bch_gen_degree(const orc_bch *b);
const int *orc_bch_gen(const orc_bch *b);
/* DVB-S2 bit order (first bit = highest-degree coefficient), Encoder_BCH_DVBS2.cpp:28-43 */
void     orc_bch_encode(const orc_bch *b, const int32_t *info_K, int32_t *cw_N);
/* Decoder_BCH_DVBS2.cpp:28-40 around Decoder_BCH_std: returns status (0 ok), cwd=!status */
int      orc_bch_decode(const orc_bch *b, const int32_t *in_N, int32_t *out_K, int8_t *cwd);

/* ---------------------------------------------------------------- modem (a3), interleaver (a4) */
/* cstl: n_pts complex points (re,im interleaved) as read from conf/mod/x.mod; normalised here
 * to unit mean energy (tools::Constellation_user).  out: n_pts*2 floats */
void orc_cstl_normalise(const float *cstl_in, int n_pts, float *cstl_out);
/* bits LSB-first per symbol */
void orc_modulate(const float *cstl, int bps, const int32_t *bits, int n_bits, float *sym);
/* exact max* demapper, L = max*_{b=0}(-d2/(2 sigma^2)) - max*_{b=1}(...) (Modem_generic, MAX=max_star) */
void orc_demodulate(const float *cstl, int bps, float sigma, const float *sym, int n_sym, float *llr);
/* column/row interleaver LUT (Interleaver_core_column_row; DVBS2.cpp:451-476).
 * order 0 = TOP_LEFT, 1 = TOP_RIGHT, n_cols <= 1 -> identity.  itl[i] = nat[lut[i]] */
void orc_itl_lut(int N, int n_cols, int order, uint32_t *lut);

/* ---------------------------------------------------------------- glue (a6-a9) */
/* Gold sequence R_n(i), n = 0 (ETSI 5.5.4); pinned against Scrambler_PL.hpp:54-4207 */
void orc_pl_rand_seq(int n, uint8_t *seq);
/* Scrambler_PL.hxx:61-78: n_sym symbols, first start_ix copied */
void orc_pl_scramble(const float *in, float *out, int n_sym, int start_ix, int scramble);
/* Scrambler_BB.hxx:51-72 */
void orc_bb_scramble(const int32_t *in, int32_t *out, int n);
/* Framer.hxx:97-196: 90-symbol PLHEADER from the 7-bit mod_cod vector */
void orc_plheader(const int *mod_cod7, float *plh180);
/* Framer.hxx:232-293 / :330-343 */
void orc_framer_generate(const float *xfec, int n_xfec_sym, const float *plh180, float *plframe);
void orc_framer_remove_plh(const float *plframe, int n_xfec_sym, float *xfec);
int  orc_pl_frame_size(int n_xfec_sym);
/* Estimator_DVBS2.hxx:31-58; out[0]=sigma out[1]=ebn0 out[2]=esn0 */
void orc_estimate(const float *xfec, int n_sym, float code_rate, int bps, float *out3);
/* Multiplier_AGC_cc_naive.cpp:22-46: one frame of n_cplx complex samples over its standard deviation about its mean, to `output_energy` (float sums in order, as there) */
void orc_agc(const float *x, int n_cplx, float output_energy, float *z);
/* Multiplier_sine_ccc_naive.cpp:44-51,69-77 (set_nu, step): n_cplx samples of the stream from position *n on (updated), float throughout as in the module's <float> form */
void orc_nco(const float *x, int n_cplx, float nu, float *n, float *z);
/* Filter_RRC_ccr_naive.cpp:13-48 */
void orc_rrc_taps(float rolloff, int osf, int grp_delay, float *taps);
/* Filter_FIR_ccr.cpp:68-142 + .hpp:39-52: streaming FIR, hist = last (T-1) complex samples
 * of the previous frame (updated in place). n complex samples. */
void orc_fir(const float *taps, int T, float *hist, const float *x, float *y, int n);
/* Filter_UPFIR_ccr_naive.cpp:52-66 (row N2): zero-stuff by osf then FIR */
void orc_upfir(const float *taps, int T, int osf, float *hist, const float *x, float *y, int n_in);

/* ---------------------------------------------------------------- frame synchronizer (row N4) */
/* Variable_delay_cc_naive (Filter/Variable_delay/Variable_delay_cc_naive.cpp:14-69): the block form
 * `_filter`, including what it does while the delay changes (it re-reads the output buffer of the
 * previous call, which the caller therefore has to pass again). N floats per call. */
typedef struct orc_vdelay orc_vdelay;
orc_vdelay *orc_vdelay_create(int N, int delay, int max_delay);
void        orc_vdelay_destroy(orc_vdelay *d);
void        orc_vdelay_set_delay(orc_vdelay *d, int delay);
void        orc_vdelay_reset(orc_vdelay *d);
void        orc_vdelay_filter(orc_vdelay *d, const float *X, float *Y);
/* Synchronizer_frame_DVBS2_fast (Synchronizer_frame_DVBS2_fast.cpp/.hpp), n_frames = 1 per call.
 * n_cplx = PL frame length in symbols (the factory passes N = 2 * pl_frame, Synchronizer_frame.cpp:72);
 * vec_width = mipp::N<float>() of the reference build (8 on AVX2, 16 on AVX-512, 4 on SSE/NEON): the
 * samples past the last full vector skip the alpha-averaging (.cpp:113-114 / :284-285). */
typedef struct orc_sfm orc_sfm;
orc_sfm *orc_sfm_create(int n_cplx, float alpha, float trigger, int vec_width);
void     orc_sfm_destroy(orc_sfm *s);
void     orc_sfm_reset(orc_sfm *s);                                                        /* .cpp:304-318 */
void     orc_sfm_synchronize1(orc_sfm *s, const float *X, float *cor_SOF, float *cor_PLSC); /* .cpp:132-150 */
/* .cpp:222-299; Y is read as well as written (see orc_vdelay_filter); returns the delay */
int      orc_sfm_synchronize2(orc_sfm *s, const float *X, const float *cor_SOF, const float *cor_PLSC, float *Y);
int      orc_sfm_synchronize(orc_sfm *s, const float *X, float *Y);                        /* .cpp:46-128 */
float    orc_sfm_metric(const orc_sfm *s);                                                 /* _get_metric, .hpp:59 */
int      orc_sfm_packet_flag(const orc_sfm *s);                                            /* _get_packet_flag, .hpp:60 */
void     orc_sfm_taps(const float **sof, int *n_sof, const float **plsc, int *n_plsc);     /* conj_SOF / conj_PLSC, .hpp:19-33 */

/* ---------------------------------------------------------------- fine frequency / phase synchronizers (row N4) */
/* pilot positions of both: 1530 + 1476 k < n_cplx (Synchronizer_Luise_Reggiannini_DVBS2_aib.cpp:18-24) */
int  orc_sff_pilots(int n_cplx, int *pilot_start, int max);
/* Synchronizer_Luise_Reggiannini_DVBS2_aib::_synchronize (.cpp:93-167); R_l[2] = the damped autocorrelation
 * carried between frames (zero after reset); returns estimated_freq (the FRQ socket; PHS stays 0) */
float orc_lr_synchronize(int n_cplx, float alpha, float *R_l, const float *X, float *Y);
/* Synchronizer_freq_phase_DVBS2_aib::_synchronize (.cpp:44-112), stateless; out2 = {estimated_freq, estimated_phase} */
void orc_fp_synchronize(int n_cplx, const float *X, float *Y, float *out2);

/* ---------------------------------------------------------------- CPU baseline leg of bench.py */
/* decode F frames with `threads` threads (frames sharded); returns seconds */
double orc_ldpc_decode_batch(const orc_ldpc *c, const float *llr, int F, int sched, int n_ite,
                             float alpha, int32_t *bits, int threads);
/* inter-frame SIMD flavour (`--dec-simd INTER` of the reference: one frame per lane of a vector register, 16 with AVX-512, 8 with
 * AVX2 -- orc_ldpc_inter_width() says which this build uses): NMS, natural row order, fixed n_ite, bit-identical to the scalar
 * decoder; returns seconds */
int orc_ldpc_inter_width(void);
double orc_ldpc_decode_batch_inter(const orc_ldpc *c, const float *llr, int F, int n_ite, float alpha,
                                   int32_t *bits, int threads);
/* the CPU baseline's form (bench.py): threads pinned to the caller's CPU list, work buffers and a private copy of each thread's share of the LLRs first-touched by the
 * thread itself, static deal of the blocks; flavour 0 = scalar (natural order), 1 = inter-frame SIMD; returns the seconds between the barriers around the decode */
double orc_ldpc_decode_batch_pinned(const orc_ldpc *c, const float *llr, int F, int flavour, int n_ite, float alpha, int32_t *bits, int threads,
                                    const int *cpus, int n_cpus, double *per_thread_max, double *per_thread_min);
/* STREAM triad on pinned threads over first-touched arrays: sustainable DRAM GB/s of the host (3 x 4 bytes per element) */
double orc_stream_triad_GBps(int threads, size_t floats_per_thread, int reps, const int *cpus, int n_cpus);
#ifdef __cplusplus
}
#endif
#endif
