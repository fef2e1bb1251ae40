/*
 * dvbs2_oracle.c -- CPU ORACLE for the DVB-S2 RX inner path.  TEST INFRASTRUCTURE ONLY
 * (see dvbs2_oracle.h for the rules and the parity-pinning status: a1-a4 "parity
 * unpinned", the reference's arithmetic lives in the absent lib/aff3ct).
 *
 * Plain C restatement, scalar, fp32 where the reference is fp32 (R = Q = float, B = int).
 * Compiled with -ffp-contract=off so the fp32 results do not depend on FMA fusion.
 * All path:line citations are relative to /root/reference.
 */
#define _GNU_SOURCE      /* sched_setaffinity, CPU_SET (the CPU baseline's pinned threads) */
#include "dvbs2_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>
#ifdef _OPENMP
#include <sched.h>
#include <omp.h>
#endif

/* ======================================================================== LDPC */
struct orc_ldpc {
    int N, K, M, q, n_rows, E;
    int *row_ptr, *addr;         /* ETSI address table */
    int *chk_ptr, *chk_var;      /* H by check, natural order; per check: info vars in
                                    (table row, position) order, then p_c, then p_{c-1} */
    int *chk_lvl;                /* per edge: QC-schedule conflict level (0 = primary) */
    int max_deg;
};

static inline int f_signbit(float x) { uint32_t u; memcpy(&u, &x, 4); return (int)(u >> 31); }

orc_ldpc *orc_ldpc_create(int N, int K, int n_rows, const int *row_ptr, const int *addr)
{
    orc_ldpc *c = (orc_ldpc *)calloc(1, sizeof *c);
    c->N = N; c->K = K; c->M = N - K; c->q = c->M / 360; c->n_rows = n_rows;
    if (c->M % 360 || K % 360 || n_rows != K / 360) { free(c); return NULL; }
    c->row_ptr = (int *)malloc(sizeof(int) * (n_rows + 1));
    memcpy(c->row_ptr, row_ptr, sizeof(int) * (n_rows + 1));
    int na = row_ptr[n_rows];
    c->addr = (int *)malloc(sizeof(int) * na);
    memcpy(c->addr, addr, sizeof(int) * na);
    const int M = c->M, q = c->q;
    /* check degrees */
    int *deg = (int *)calloc(M, sizeof(int));
    for (int g = 0; g < n_rows; g++)
        for (int p = row_ptr[g]; p < row_ptr[g + 1]; p++)
            for (int m = 0; m < 360; m++) deg[(addr[p] + m * q) % M]++;
    for (int k = 0; k < M; k++) deg[k] += (k == 0) ? 1 : 2;
    c->chk_ptr = (int *)malloc(sizeof(int) * (M + 1));
    c->chk_ptr[0] = 0;
    for (int k = 0; k < M; k++) {
        c->chk_ptr[k + 1] = c->chk_ptr[k] + deg[k];
        if (deg[k] > c->max_deg) c->max_deg = deg[k];
    }
    c->E = c->chk_ptr[M];
    c->chk_var = (int *)malloc(sizeof(int) * c->E);
    c->chk_lvl = (int *)calloc(c->E, sizeof(int));
    int *fill = (int *)calloc(M, sizeof(int));
    /* ETSI EN 302 307 5.3.2: info bit i = 360 g + m accumulates into parity address
     * (x + m q) mod (N-K) for every x in table row g */
    for (int g = 0; g < n_rows; g++)
        for (int p = row_ptr[g]; p < row_ptr[g + 1]; p++)
            for (int m = 0; m < 360; m++) {
                int k = (addr[p] + m * q) % M;
                c->chk_var[c->chk_ptr[k] + fill[k]++] = g * 360 + m;
            }
    /* p_k = p_k xor p_{k-1}: check k holds parity bits k and k-1 */
    for (int k = 0; k < M; k++) {
        c->chk_var[c->chk_ptr[k] + fill[k]++] = K + k;
        if (k > 0) c->chk_var[c->chk_ptr[k] + fill[k]++] = K + k - 1;
    }
    /* conflict levels for the QC schedule: within one check, an info edge whose bit-GROUP
     * already appeared earlier in the same check's slot list is a "later" edge */
    for (int k = 0; k < M; k++) {
        int b = c->chk_ptr[k], e = c->chk_ptr[k + 1];
        for (int i = b; i < e; i++) {
            int v = c->chk_var[i];
            if (v >= K) continue;
            int lvl = 0;
            for (int j = b; j < i; j++)
                if (c->chk_var[j] < K && c->chk_var[j] / 360 == v / 360) lvl++;
            c->chk_lvl[i] = lvl;
        }
    }
    free(fill); free(deg);
    return c;
}

void orc_ldpc_destroy(orc_ldpc *c)
{
    if (!c) return;
    free(c->row_ptr); free(c->addr); free(c->chk_ptr); free(c->chk_var); free(c->chk_lvl); free(c);
}
int orc_ldpc_n_edges(const orc_ldpc *c) { return c->E; }
int orc_ldpc_q(const orc_ldpc *c) { return c->q; }
void orc_ldpc_csr(const orc_ldpc *c, const int **chk_ptr, const int **chk_var)
{ *chk_ptr = c->chk_ptr; *chk_var = c->chk_var; }

void orc_ldpc_encode(const orc_ldpc *c, const int32_t *info, int32_t *cw)
{
    const int K = c->K, M = c->M, q = c->q;
    memcpy(cw, info, sizeof(int32_t) * K);
    int32_t *p = cw + K;
    memset(p, 0, sizeof(int32_t) * M);
    for (int g = 0; g < c->n_rows; g++)
        for (int m = 0; m < 360; m++) {
            if (!(info[g * 360 + m] & 1)) continue;
            for (int a = c->row_ptr[g]; a < c->row_ptr[g + 1]; a++)
                p[(c->addr[a] + m * q) % M] ^= 1;
        }
    for (int k = 1; k < M; k++) p[k] ^= p[k - 1];
}

int orc_ldpc_syndrome_weight(const orc_ldpc *c, const int32_t *cw)
{
    int w = 0;
    for (int k = 0; k < c->M; k++) {
        int s = 0;
        for (int i = c->chk_ptr[k]; i < c->chk_ptr[k + 1]; i++) s ^= cw[c->chk_var[i]] & 1;
        w += s;
    }
    return w;
}

static int soft_syndrome_ok(const orc_ldpc *c, const float *L)
{
    for (int k = 0; k < c->M; k++) {
        int s = 0;
        for (int i = c->chk_ptr[k]; i < c->chk_ptr[k + 1]; i++) s ^= (L[c->chk_var[i]] < 0.0f);
        if (s) return 0;
    }
    return 1;
}

/* one check-node update.  v2c[] in, c2v_new[] out. */
static void chk_update_nms(const float *v2c, int d, float alpha, float *out)
{
    float min1 = FLT_MAX, min2 = FLT_MAX;
    int sign = 0;
    for (int j = 0; j < d; j++) {
        float a = fabsf(v2c[j]);
        sign ^= f_signbit(v2c[j]);
        float t = a > min1 ? a : min1;           /* max(a, min1) */
        min2 = min2 < t ? min2 : t;
        min1 = min1 < a ? min1 : a;
    }
    float cst1 = min2 * alpha, cst2 = min1 * alpha;
    for (int j = 0; j < d; j++) {
        float a = fabsf(v2c[j]);
        float mag = (a == min1) ? cst1 : cst2;
        int s = sign ^ f_signbit(v2c[j]);
        out[j] = s ? -mag : mag;
    }
}

/* exact sum-product check node: c->v_j = boxplus of all the OTHER v->c, evaluated with forward and
 * backward boxplus recursions in fp32:
 *     a [+] b = sign(a) sign(b) min(|a|,|b|) + log1p(exp(-|a+b|)) - log1p(exp(-|a-b|))
 * (numerically equal to 2 atanh(tanh(a/2) tanh(b/2)), the form AFF3CT's Update_rule_SPA uses, without
 * the saturation of tanh in fp32).  +inf is the neutral element. */
static inline float boxplus(float a, float b)
{
    if (a == INFINITY) return b;
    if (b == INFINITY) return a;
    const float mn = fminf(fabsf(a), fabsf(b));
    const float sg = (f_signbit(a) ^ f_signbit(b)) ? -mn : mn;
    return sg + (log1pf(expf(-fabsf(a + b))) - log1pf(expf(-fabsf(a - b))));
}
static void chk_update_spa(const float *v2c, int d, float *out)
{
    float fw[64], bw[64];
    fw[0] = INFINITY;
    for (int j = 1; j < d; j++) fw[j] = boxplus(fw[j - 1], v2c[j - 1]);
    bw[d - 1] = INFINITY;
    for (int j = d - 2; j >= 0; j--) bw[j] = boxplus(bw[j + 1], v2c[j + 1]);
    for (int j = 0; j < d; j++) out[j] = boxplus(fw[j], bw[j]);
}

/* ---- the check node as AFF3CT's Update_rule_SPA evaluates it [UPSTREAM-RECALL: lib/aff3ct is an empty submodule; the rule the reference
 * selects with `--dec-implem SPA`, its default: DVBS2.cpp:135,418-449; every refs/TX_RX_BB trace says "LDPC implem = SPA", "LDPC simd = "]:
 *     in :  t_j = tanh(|v_j| / 2) in fp32 (exactly 1.0f beyond |v_j| ~ 18),  P = t_0 t_1 .. t_{d-1} multiplied in fp32 in edge order,
 *           sign = xor of the sign bits
 *     out:  val = P / t_j ;  val = (val < 1) ? val : 1 - FLT_EPSILON  (a NaN from 0 / 0 takes the second branch) ;
 *           |c->v_j| = 2 atanh(val) ;  sign = (all signs) ^ (own sign)
 * Unlike the exact boxplus above this SATURATES: no message exceeds 2 atanh(1 - 2^-23) = 16.64, and near that cap the quotient moves in steps of
 * 2^-24, i.e. the message in steps of 0.1 .. 0.7.  Because of those steps a faithful GPU twin has to agree BIT FOR BIT, not to 1e-4: every operation
 * below is one whose IEEE-754 result is correctly rounded (add, multiply, fma, divide, round-to-nearest-integer) -- tanh and log1p are written out with those
 * (the structure of the usual libm routines: tanh through expm1, 2 atanh(x) = log1p(2 x / (1 - x))) instead of calling libm, whose results differ from
 * one library to the next by an ulp.  tests/test_oracle.py holds them to libm within 4 ulp.  Edge order = the order the checks' edges are built in above:
 * information bits in the order of the standard's address table, then p_k, then p_(k-1). */
static inline float det_scale2(int n) { uint32_t u = (uint32_t)(n + 127) << 23; float f; memcpy(&f, &u, 4); return f; }     /* 2^n, -126 <= n <= 127 */
/* e^y - 1, -2.1 <= y <= 45 */
static inline float det_expm1(float y)
{
    const float n = rintf(y * 1.44269502f);
    float r = fmaf(-n, 0.693145751953125f, y);              /* ln 2 = hi + lo, hi with 11 trailing zero bits: n hi is exact */
    r = fmaf(-n, 1.42860677e-6f, r);
    float q = 1.98412701e-4f;                               /* 1/5040 .. 1/2: e^r - 1 = r + r^2 (1/2 + r (1/6 + ...)), |r| <= 0.347 */
    q = fmaf(q, r, 1.38888892e-3f);
    q = fmaf(q, r, 8.33333377e-3f);
    q = fmaf(q, r, 4.16666679e-2f);
    q = fmaf(q, r, 1.66666672e-1f);
    q = fmaf(q, r, 0.5f);
    const float pm1 = fmaf(q * r, r, r);
    const float sc = det_scale2((int)n);
    return fmaf(sc, pm1, sc - 1.0f);
}
/* tanh(a / 2), a >= 0 (+inf allowed) */
static inline float det_tanh_half(float a)
{
    if (!(a < 44.0f)) return 1.0f;
    if (a >= 2.0f) { const float t = det_expm1(a); return 1.0f - 2.0f / (t + 2.0f); }
    const float t = det_expm1(-a);
    return -t / (t + 2.0f);
}
/* log(1 + w), 0 <= w < 2^26 */
static inline float det_log1p(float w)
{
    const float u = 1.0f + w;
    const float c = w - (u - 1.0f);                         /* what the rounding of 1 + w lost (u - 1 is exact) */
    uint32_t iu; memcpy(&iu, &u, 4);
    int e = (int)(iu >> 23) - 127;
    uint32_t im = (iu & 0x007FFFFFu) | 0x3F800000u;         /* mantissa in [1, 2) */
    if (im >= 0x3FB504F3u) { im -= 0x00800000u; e += 1; }   /* ... in [sqrt(1/2), sqrt 2) */
    float m; memcpy(&m, &im, 4);
    const float f = fmaf(c, det_scale2(-e), m - 1.0f);      /* 1 + w = 2^e (m + c 2^-e): the lost part goes back in at the mantissa's scale (no second division) */
    const float s = f / (2.0f + f);
    const float z = s * s;
    float q = 0.111111112f;                                 /* log m = 2 atanh(s) = 2 s (1 + z / 3 + z^2 / 5 + z^3 / 7 + z^4 / 9), z <= 0.0295 */
    q = fmaf(q, z, 0.142857149f);
    q = fmaf(q, z, 0.2f);
    q = fmaf(q, z, 0.333333343f);
    const float s2 = s + s;
    const float lm = fmaf(s2 * z, q, s2);
    const float fe = (float)e;
    float r = fmaf(fe, 0.693145751953125f, lm);
    r = fmaf(fe, 1.42860677e-6f, r);
    return r;
}
static void chk_update_spa_tanh(const float *v2c, int d, float *out)
{
    float t[64], P = 1.0f;
    int sign = 0;
    for (int j = 0; j < d; j++) {
        t[j] = det_tanh_half(fabsf(v2c[j]));
        sign ^= f_signbit(v2c[j]);
        P = P * t[j];
    }
    for (int j = 0; j < d; j++) {
        float val = P / t[j];
        val = (val < 1.0f) ? val : 1.0f - FLT_EPSILON;
        const float mag = det_log1p((val + val) / (1.0f - val));
        out[j] = (sign ^ f_signbit(v2c[j])) ? -mag : mag;
    }
}
/* test hooks: the two written-out functions against libm */
float orc_det_tanh_half(float a) { return det_tanh_half(a); }
float orc_det_log1p(float w) { return det_log1p(w); }
static inline void chk_update(int implem, float alpha, const float *v2c, int d, float *out)
{
    if (implem == ORC_NMS) chk_update_nms(v2c, d, alpha, out);
    else if (implem == ORC_SPA_TANH) chk_update_spa_tanh(v2c, d, out);
    else if (implem == ORC_SPA_CLIP) {      /* the exact rule, every message clipped to the cap of the tanh-product rule: 2 atanh(1 - FLT_EPSILON) */
        chk_update_spa(v2c, d, out);
        for (int j = 0; j < d; j++) if (fabsf(out[j]) > 16.6355324f) out[j] = copysignf(16.6355324f, out[j]);
    }
    else chk_update_spa(v2c, d, out);
}
void orc_chk_update(int implem, float alpha, const float *v2c, int d, float *out) { chk_update(implem, alpha, v2c, d, out); }

/* work buffers of one decoder instance (a thread of the CPU baseline keeps its own across frames:
 * allocating ~1 MB per frame from 256 threads at once serialises them in the allocator) */
typedef struct { float *L, *msg, *v2c, *nw; } ldpc_ws;
static void ldpc_ws_alloc(const orc_ldpc *c, ldpc_ws *w)
{
    w->L = (float *)malloc(sizeof(float) * c->N);
    w->msg = (float *)malloc(sizeof(float) * c->E);
    w->v2c = (float *)malloc(sizeof(float) * 360 * c->max_deg);
    w->nw = (float *)malloc(sizeof(float) * 360 * c->max_deg);
}
static void ldpc_ws_free(ldpc_ws *w) { free(w->L); free(w->msg); free(w->v2c); free(w->nw); }

static int ldpc_decode_ws(const orc_ldpc *c, const float *llr, int implem, int sched, int n_ite,
                          float alpha, int early_stop, int32_t *bits_K, float *post, int8_t *cwd, ldpc_ws *ws)
{
    const int N = c->N, M = c->M, q = c->q;
    float *L = ws->L, *msg = ws->msg, *v2c = ws->v2c, *nw = ws->nw;
    memset(msg, 0, sizeof(float) * c->E);                   /* c->v, zero-initialised */
    memcpy(L, llr, sizeof(float) * N);
    int ite = 0;
    int max_lvl = 0;
    for (int i = 0; i < c->E; i++) if (c->chk_lvl[i] > max_lvl) max_lvl = c->chk_lvl[i];
    for (; ite < n_ite; ) {
        if (sched == ORC_SCHED_NATURAL || sched == ORC_SCHED_QC_SEQ) {
            /* ORC_SCHED_QC_SEQ (an analysis aid, round 6): the checks in the ORDER of the QC layers (layer r = checks q t + r, t = 0 .. 359) but one after the other, every
             * check reading what the one before it wrote -- separates what the QC schedule's ORDER costs against the row order from what its 360-check SNAPSHOT costs */
            for (int kk = 0; kk < M; kk++) {
                const int k = sched == ORC_SCHED_NATURAL ? kk : q * (kk % 360) + kk / 360;
                int b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                for (int j = 0; j < d; j++) v2c[j] = L[c->chk_var[b + j]] - msg[b + j];
                chk_update(implem, alpha, v2c, d, nw);
                for (int j = 0; j < d; j++) { msg[b + j] = nw[j]; L[c->chk_var[b + j]] = v2c[j] + nw[j]; }
            }
        } else {
            /* QC-layer schedule: layer r = checks {q t + r, t = 0..359}; all 360 checks of a
             * layer read their posteriors before any of them writes (phase 1); primary
             * edges write v2c + new (phase 2); level-k duplicate edges then add
             * (new - old) in level order (phase 3). */
            const int D = c->max_deg;
            float *acc = sched == ORC_SCHED_QC_FIX ? (float *)calloc((size_t)N, sizeof(float)) : NULL;
            for (int r = 0; r < q; r++) {
                for (int t = 0; t < 360; t++) {
                    int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                    for (int j = 0; j < d; j++) v2c[t * D + j] = L[c->chk_var[b + j]] - msg[b + j];
                    chk_update(implem, alpha, v2c + t * D, d, nw + t * D);
                }
                if (sched == ORC_SCHED_QC_FIX) {
                    /* ORC_SCHED_QC_FIX (an analysis aid, round 6): one Jacobi correction inside the layer -- a check whose bit is shared with another check of the layer
                     * recomputes with that bit's value moved by the other check's first-pass delta (what a second, parallel pass over the layer's checks would cost) */
                    for (int t = 0; t < 360; t++) {
                        int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                        for (int j = 0; j < d; j++) acc[c->chk_var[b + j]] += nw[t * D + j] - msg[b + j];
                    }
                    for (int t = 0; t < 360; t++) {
                        int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                        for (int j = 0; j < d; j++) v2c[t * D + j] += acc[c->chk_var[b + j]] - (nw[t * D + j] - msg[b + j]);      /* the OTHER checks' deltas on this bit */
                    }
                    for (int t = 0; t < 360; t++) {
                        int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                        float tmp[64];
                        chk_update(implem, alpha, v2c + t * D, d, tmp);
                        for (int j = 0; j < d; j++) { acc[c->chk_var[b + j]] = 0.f; v2c[t * D + j] = L[c->chk_var[b + j]] - msg[b + j]; nw[t * D + j] = tmp[j]; }
                    }
                }
                for (int t = 0; t < 360; t++) {
                    int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                    for (int j = 0; j < d; j++)
                        if (c->chk_lvl[b + j] == 0) L[c->chk_var[b + j]] = v2c[t * D + j] + nw[t * D + j];
                }
                for (int lvl = 1; lvl <= max_lvl; lvl++)
                    for (int t = 0; t < 360; t++) {
                        int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                        for (int j = 0; j < d; j++)
                            if (c->chk_lvl[b + j] == lvl) {
                                float delta = nw[t * D + j] - msg[b + j];
                                L[c->chk_var[b + j]] = L[c->chk_var[b + j]] + delta;
                            }
                    }
                for (int t = 0; t < 360; t++) {
                    int k = q * t + r, b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
                    for (int j = 0; j < d; j++) msg[b + j] = nw[t * D + j];
                }
            }
            free(acc);
        }
        ite++;
        if (early_stop && soft_syndrome_ok(c, L)) break;
    }
    if (cwd) *cwd = (int8_t)soft_syndrome_ok(c, L);
    if (bits_K) for (int i = 0; i < c->K; i++) bits_K[i] = L[i] < 0.0f;
    if (post) memcpy(post, L, sizeof(float) * N);
    return ite;
}

int orc_ldpc_decode(const orc_ldpc *c, const float *llr, int implem, int sched, int n_ite,
                    float alpha, int early_stop, int32_t *bits_K, float *post, int8_t *cwd)
{
    ldpc_ws ws;
    ldpc_ws_alloc(c, &ws);
    const int ite = ldpc_decode_ws(c, llr, implem, sched, n_ite, alpha, early_stop, bits_K, post, cwd, &ws);
    ldpc_ws_free(&ws);
    return ite;
}

double orc_ldpc_decode_batch(const orc_ldpc *c, const float *llr, int F, int sched, int n_ite,
                             float alpha, int32_t *bits, int threads)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        ldpc_ws ws;
        ldpc_ws_alloc(c, &ws);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int f = 0; f < F; f++)
            ldpc_decode_ws(c, llr + (size_t)f * c->N, ORC_NMS, sched, n_ite, alpha, 0,
                           bits ? bits + (size_t)f * c->K : NULL, NULL, NULL, &ws);
        ldpc_ws_free(&ws);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    (void)threads;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- inter-frame SIMD flavour of the CPU baseline: what `--dec-simd INTER` does in the reference
 * (README.md:171-178): W frames are decoded together, frame index fastest in memory, every
 * per-edge operation is one vector operation across the W frames (GCC vector types: AVX2 or
 * AVX-512 registers, whichever the build's -march has).  Natural row order, NMS, fixed n_ite.  Same arithmetic as
 * orc_ldpc_decode(.., ORC_NMS, ORC_SCHED_NATURAL, ..): results are bit-identical (tests). */
#ifdef __AVX512F__
#define ORC_W 16       /* frames per vector = floats per register, as mipp::N<float>() in the reference */
#else
#define ORC_W 8
#endif
typedef struct { float *L, *msg, *v2c; } inter_ws;       /* one per thread, reused across blocks */
static void inter_ws_alloc(const orc_ldpc *c, inter_ws *w)
{
    w->L = (float *)aligned_alloc(64, sizeof(float) * (size_t)c->N * ORC_W);
    w->msg = (float *)aligned_alloc(64, sizeof(float) * (size_t)c->E * ORC_W);
    w->v2c = (float *)aligned_alloc(64, sizeof(float) * (size_t)c->max_deg * ORC_W);
}
static void inter_ws_free(inter_ws *w) { free(w->L); free(w->msg); free(w->v2c); }

/* one vector = the ORC_W frames of a block (GCC vector extensions: one zmm register with AVX-512, two ymm with AVX2); the running
 * minima and the sign word stay in registers across a check's edges */
typedef float vf32 __attribute__((vector_size(4 * ORC_W), aligned(4 * ORC_W)));       /* (aligned to its own size: the arrays are indexed at 4 ORC_W-byte strides, 32 bytes in the portable AVX2 build) */
typedef int32_t vi32 __attribute__((vector_size(4 * ORC_W), aligned(4 * ORC_W)));
#define v_sel(m, a, b) ((vf32)(((vi32)(a) & (m)) | ((vi32)(b) & ~(m))))      /* m ? a : b, lane by lane (a macro: a 64-byte vector argument would go through memory in an AVX2 build) */

static __attribute__((noinline)) void decode_inter_block(const orc_ldpc *c, const float *llr, int nf, int n_ite, float alpha, int32_t *bits, inter_ws *ws)
{
    const int N = c->N, M = c->M;
    float *L = ws->L, *msg = ws->msg;
    vf32 *v2c = (vf32 *)ws->v2c;
    const vi32 absm = (vi32){0} + 0x7FFFFFFF, sgnm = (vi32){0} + (int32_t)0x80000000u;
    const vf32 fmax = (vf32){0} + FLT_MAX, va = (vf32){0} + alpha;
    memset(msg, 0, sizeof(float) * (size_t)c->E * ORC_W);
    for (int v = 0; v < N; v++)
        for (int w = 0; w < ORC_W; w++) L[(size_t)v * ORC_W + w] = w < nf ? llr[(size_t)w * N + v] : 1.0f;
    for (int it = 0; it < n_ite; it++)
        for (int k = 0; k < M; k++) {
            const int b = c->chk_ptr[k], d = c->chk_ptr[k + 1] - b;
            vf32 min1 = fmax, min2 = fmax;
            vi32 sg = (vi32){0};
            for (int j = 0; j < d; j++) {
                const vf32 x = *(const vf32 *)(L + (size_t)c->chk_var[b + j] * ORC_W) - *(const vf32 *)(msg + (size_t)(b + j) * ORC_W);
                v2c[j] = x;
                const vf32 a = (vf32)((vi32)x & absm);
                sg ^= (vi32)x;
                const vf32 t = v_sel(a > min1, a, min1);          /* max(a, min1) */
                min2 = v_sel(min2 < t, min2, t);
                min1 = v_sel(min1 < a, min1, a);
            }
            const vf32 cst1 = min2 * va, cst2 = min1 * va;
            for (int j = 0; j < d; j++) {
                const vf32 x = v2c[j];
                const vf32 a = (vf32)((vi32)x & absm);
                const vf32 mag = v_sel(a == min1, cst1, cst2);
                const vf32 nw = (vf32)((vi32)mag | ((sg ^ (vi32)x) & sgnm));
                *(vf32 *)(msg + (size_t)(b + j) * ORC_W) = nw;
                *(vf32 *)(L + (size_t)c->chk_var[b + j] * ORC_W) = x + nw;
            }
        }
    if (bits)
        for (int w = 0; w < nf; w++)
            for (int i = 0; i < c->K; i++) bits[(size_t)w * c->K + i] = L[(size_t)i * ORC_W + w] < 0.0f;
}

int orc_ldpc_inter_width(void) { return ORC_W; }

double orc_ldpc_decode_batch_inter(const orc_ldpc *c, const float *llr, int F, int n_ite, float alpha, int32_t *bits, int threads)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int nb = (F + ORC_W - 1) / ORC_W;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        inter_ws ws;
        inter_ws_alloc(c, &ws);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int blk = 0; blk < nb; blk++) {
            const int f0 = blk * ORC_W, nf = F - f0 < ORC_W ? F - f0 : ORC_W;
            decode_inter_block(c, llr + (size_t)f0 * c->N, nf, n_ite, alpha, bits ? bits + (size_t)f0 * c->K : NULL, &ws);
        }
        inter_ws_free(&ws);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    (void)threads;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- CPU baseline, the form bench.py times (round 5): every thread pinned to a CPU of the caller's list, its work buffers AND a private copy of its share of the
 * LLRs first-touched by itself (NUMA-local pages), the blocks dealt statically (thread t takes blocks t, t + T, ..); the timed region starts behind a barrier, when every
 * copy is in place, and ends when the slowest thread is done.  flavour 0: scalar decoder (sched = natural), 1: inter-frame SIMD.  Same arithmetic as the functions above.
 * Returns seconds; *per_thread_max / *per_thread_min (may be NULL) get the slowest / fastest thread's own time. */
double orc_ldpc_decode_batch_pinned(const orc_ldpc *c, const float *llr, int F, int flavour, int n_ite, float alpha, int32_t *bits, int threads,
                                    const int *cpus, int n_cpus, double *tmax, double *tmin)
{
    const int W = flavour ? ORC_W : 1, nb = (F + W - 1) / W;
    double t_start = 0.0, t_end = 0.0, hi = 0.0, lo = 1e30;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int tid = 0, T = 1;
#endif
        cpu_set_t old, one;
        int pinned = 0;
        if (cpus && n_cpus > 0 && sched_getaffinity(0, sizeof old, &old) == 0) {
            CPU_ZERO(&one); CPU_SET(cpus[tid % n_cpus], &one);
            pinned = sched_setaffinity(0, sizeof one, &one) == 0;
        }
        inter_ws iw = {0, 0, 0};
        ldpc_ws sw = {0, 0, 0, 0};
        if (flavour) inter_ws_alloc(c, &iw); else ldpc_ws_alloc(c, &sw);
        int mine = 0;
        for (int blk = tid; blk < nb; blk += T) mine++;
        float *own = (float *)malloc(sizeof(float) * (size_t)(mine > 0 ? mine : 1) * W * c->N);
        int k = 0;
        for (int blk = tid; blk < nb; blk += T, k++) {
            const int f0 = blk * W, nf = F - f0 < W ? F - f0 : W;
            memcpy(own + (size_t)k * W * c->N, llr + (size_t)f0 * c->N, sizeof(float) * (size_t)nf * c->N);
        }
        if (flavour) { memset(iw.L, 0, sizeof(float) * (size_t)c->N * ORC_W); memset(iw.msg, 0, sizeof(float) * (size_t)c->E * ORC_W); }      /* pages touched before the clock starts */
        else { memset(sw.L, 0, sizeof(float) * c->N); memset(sw.msg, 0, sizeof(float) * c->E); }
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
        { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); t_start = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
#ifdef _OPENMP
#pragma omp barrier
#endif
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        k = 0;
        for (int blk = tid; blk < nb; blk += T, k++) {
            const int f0 = blk * W, nf = F - f0 < W ? F - f0 : W;
            if (flavour) decode_inter_block(c, own + (size_t)k * W * c->N, nf, n_ite, alpha, bits ? bits + (size_t)f0 * c->K : NULL, &iw);
            else ldpc_decode_ws(c, own + (size_t)k * c->N, ORC_NMS, ORC_SCHED_NATURAL, n_ite, alpha, 0, bits ? bits + (size_t)f0 * c->K : NULL, NULL, NULL, &sw);
        }
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double mine_s = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (mine > 0 && mine_s > hi) hi = mine_s; if (mine > 0 && mine_s < lo) lo = mine_s; }
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
        { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); t_end = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
        free(own);
        if (flavour) inter_ws_free(&iw); else ldpc_ws_free(&sw);
        if (pinned) sched_setaffinity(0, sizeof old, &old);
    }
    if (tmax) *tmax = hi;
    if (tmin) *tmin = lo;
    return t_end - t_start;
}

/* STREAM triad a[i] = b[i] + s c[i] on `threads` pinned threads, every thread on arrays it first-touched itself: the host's sustainable DRAM bandwidth in GB/s
 * (3 x 4 bytes per element; write-allocate traffic not counted, as STREAM does), best of `reps` passes.  Beside the CPU baseline it says whether the inter-frame
 * flavour -- 16 frames x (N + E) floats = 16.6 MB of state per thread, swept once per iteration -- is bound by the cores or by the memory they share. */
double orc_stream_triad_GBps(int threads, size_t floats_per_thread, int reps, const int *cpus, int n_cpus)
{
    double best = 0.0;
    if (threads < 1) threads = 1;
    double t0 = 0.0, t1 = 0.0;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int tid = 0, T = 1;
#endif
        cpu_set_t old, one;
        int pinned = 0;
        if (cpus && n_cpus > 0 && sched_getaffinity(0, sizeof old, &old) == 0) {
            CPU_ZERO(&one); CPU_SET(cpus[tid % n_cpus], &one);
            pinned = sched_setaffinity(0, sizeof one, &one) == 0;
        }
        float *a = (float *)aligned_alloc(64, sizeof(float) * floats_per_thread), *b = (float *)aligned_alloc(64, sizeof(float) * floats_per_thread),
              *cc = (float *)aligned_alloc(64, sizeof(float) * floats_per_thread);
        for (size_t i = 0; i < floats_per_thread; i++) { a[i] = 0.f; b[i] = 1.f; cc[i] = 2.f; }
        for (int r = 0; r < reps; r++) {
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
            { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); t0 = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
#ifdef _OPENMP
#pragma omp barrier
#endif
            const float sc = 3.0f + (float)r;
            for (size_t i = 0; i < floats_per_thread; i++) a[i] = b[i] + sc * cc[i];
            __asm__ volatile("" :: "r"(a) : "memory");
#ifdef _OPENMP
#pragma omp barrier
#pragma omp master
#endif
            {
                struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); t1 = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
                const double g = 12.0 * (double)floats_per_thread * (double)T / (t1 - t0) / 1e9;
                if (g > best) best = g;
            }
        }
        free(a); free(b); free(cc);
        if (pinned) sched_setaffinity(0, sizeof old, &old);
    }
    return best;
}

/* ======================================================================== BCH */
struct orc_bch {
    int m, n, t, N, K, gdeg;
    int *exp_, *log_;   /* alpha_to / index_of */
    int *g;             /* generator, g[i] = coeff of x^i */
};

static int gf_mul(const orc_bch *b, int x, int y)
{ return (x && y) ? b->exp_[(b->log_[x] + b->log_[y]) % b->n] : 0; }
static int gf_div(const orc_bch *b, int x, int y)
{ return x ? b->exp_[(b->log_[x] - b->log_[y] + b->n) % b->n] : 0; }

orc_bch *orc_bch_create(int m, const int *prim, int t, int N, int K)
{
    orc_bch *b = (orc_bch *)calloc(1, sizeof *b);
    b->m = m; b->n = (1 << m) - 1; b->t = t; b->N = N; b->K = K;
    b->exp_ = (int *)malloc(sizeof(int) * (b->n + 1));
    b->log_ = (int *)malloc(sizeof(int) * (b->n + 1));
    int pm = 0; for (int i = 0; i <= m; i++) if (prim[i]) pm |= 1 << i;
    int x = 1;
    for (int i = 0; i < b->n; i++) {
        b->exp_[i] = x; b->log_[x] = i;
        x <<= 1; if (x >> m) x ^= pm;
    }
    b->exp_[b->n] = 1; b->log_[0] = -1;
    /* generator = lcm of the minimal polynomials of alpha^1..alpha^2t (cyclotomic cosets) */
    char *seen = (char *)calloc(b->n, 1);
    int *g = (int *)calloc(m * t + 2, sizeof(int));     /* GF(2) coefficients */
    int gdeg = 0; g[0] = 1;
    for (int j = 1; j <= 2 * t; j++) {
        if (seen[j % b->n]) continue;
        /* minimal polynomial of alpha^j over GF(2^m) coefficients, product of (x + alpha^e) */
        int mp[64]; int md = 0; mp[0] = 1;
        int e = j % b->n;
        do {
            seen[e] = 1;
            /* mp *= (x + alpha^e) */
            mp[md + 1] = 0;
            for (int i = md + 1; i >= 1; i--) mp[i] = mp[i - 1] ^ gf_mul(b, mp[i], b->exp_[e]);
            mp[0] = gf_mul(b, mp[0], b->exp_[e]);
            md++;
            e = (e * 2) % b->n;
        } while (e != j % b->n);
        /* g *= mp (mp has 0/1 coefficients) */
        int *ng = (int *)calloc(gdeg + md + 1, sizeof(int));
        for (int a = 0; a <= gdeg; a++) if (g[a])
            for (int c2 = 0; c2 <= md; c2++) if (mp[c2]) ng[a + c2] ^= 1;
        memcpy(g, ng, sizeof(int) * (gdeg + md + 1));
        gdeg += md; free(ng);
    }
    free(seen);
    b->g = g; b->gdeg = gdeg;
    if (gdeg != N - K) { /* caller checks orc_bch_gen_degree */ }
    return b;
}
void orc_bch_destroy(orc_bch *b) { if (!b) return; free(b->exp_); free(b->log_); free(b->g); free(b); }
int orc_bch_gen_degree(const orc_bch *b) { return b->gdeg; }
const int *orc_bch_gen(const orc_bch *b) { return b->g; }

void orc_bch_encode(const orc_bch *b, const int32_t *info, int32_t *cw)
{
    /* systematic: parity(x) = x^(n-k) u(x) mod g(x); DVB-S2 order: cw[i] = coeff of x^(N-1-i)
     * (Encoder_BCH_DVBS2.cpp:28-43 reverses in/out around the LSB-first aff3ct encoder) */
    const int r = b->gdeg, K = b->K;
    int *bb = (int *)calloc(r, sizeof(int));
    for (int i = 0; i < K; i++) {            /* first info bit = highest degree */
        int fb = (info[i] & 1) ^ bb[r - 1];
        for (int j = r - 1; j > 0; j--) bb[j] = bb[j - 1] ^ (b->g[j] & fb);
        bb[0] = b->g[0] & fb;
    }
    for (int i = 0; i < K; i++) cw[i] = info[i] & 1;
    for (int j = 0; j < r; j++) cw[K + j] = bb[r - 1 - j];
    free(bb);
}

int orc_bch_decode(const orc_bch *b, const int32_t *in, int32_t *out, int8_t *cwd)
{
    const int t = b->t, N = b->N, K = b->K, n = b->n;
    int S[64]; int any = 0;
    /* S_j = r(alpha^j), r(x) = sum_i in[i] x^(N-1-i) */
    for (int j = 1; j <= 2 * t; j++) {
        int s = 0;
        for (int i = 0; i < N; i++)
            if (in[i] & 1) s ^= b->exp_[(int)(((long long)j * (N - 1 - i)) % n)];
        S[j] = s; any |= s;
    }
    int nflip = 0, flip[64];
    int status = 0;
    if (any) {
        /* Berlekamp-Massey */
        int C[64] = {0}, B[64] = {0}, T[64];
        C[0] = 1; B[0] = 1;
        int L = 0, mm = 1, bb = 1;
        for (int k = 0; k < 2 * t; k++) {
            int d = S[k + 1];
            for (int i = 1; i <= L; i++) d ^= gf_mul(b, C[i], S[k + 1 - i]);
            if (d == 0) { mm++; }
            else {
                int coef = gf_div(b, d, bb);
                if (2 * L <= k) {
                    memcpy(T, C, sizeof C);
                    for (int i = 0; i + mm < 64; i++) C[i + mm] ^= gf_mul(b, coef, B[i]);
                    L = k + 1 - L; memcpy(B, T, sizeof B); bb = d; mm = 1;
                } else {
                    for (int i = 0; i + mm < 64; i++) C[i + mm] ^= gf_mul(b, coef, B[i]);
                    mm++;
                }
            }
        }
        if (L > t) status = 1;
        else {
            /* Chien search over the whole field: root alpha^(-d) <=> error at degree d */
            int count = 0;
            for (int d = 0; d < n && count <= L; d++) {
                int v = 0;
                for (int i = 0; i <= L; i++)
                    if (C[i]) v ^= b->exp_[(b->log_[C[i]] + (int)(((long long)i * (n - d)) % n)) % n];
                if (v == 0) { if (count < 64) flip[count] = d; count++; }
            }
            if (count == L) nflip = L; else status = 1;
        }
    }
    for (int i = 0; i < K; i++) out[i] = in[i] & 1;
    for (int f = 0; f < nflip; f++) {
        int pos = N - 1 - flip[f];            /* degree -> DVB-S2 position */
        if (pos >= 0 && pos < K) out[pos] ^= 1;   /* degrees >= N (shortened zeros) ignored */
    }
    if (cwd) *cwd = (int8_t)!status;
    return status;
}

/* ======================================================================== modem */
void orc_cstl_normalise(const float *in, int n_pts, float *out)
{
    float es = 0.f;
    for (int i = 0; i < n_pts; i++) es += in[2 * i] * in[2 * i] + in[2 * i + 1] * in[2 * i + 1];
    float s = sqrtf(es / (float)n_pts);
    for (int i = 0; i < 2 * n_pts; i++) out[i] = in[i] / s;
}

void orc_modulate(const float *cstl, int bps, const int32_t *bits, int n_bits, float *sym)
{
    int ns = n_bits / bps;
    for (int k = 0; k < ns; k++) {
        int idx = 0;
        for (int j = 0; j < bps; j++) idx += (1 << j) * (bits[k * bps + j] & 1);
        sym[2 * k] = cstl[2 * idx]; sym[2 * k + 1] = cstl[2 * idx + 1];
    }
}

static inline float max_star(float a, float b)
{
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    float mx = a > b ? a : b;
    return mx + log1pf(expf(-fabsf(a - b)));
}

void orc_demodulate(const float *cstl, int bps, float sigma, const float *sym, int n_sym, float *llr)
{
    const int P = 1 << bps;
    const float inv = 1.0f / (2.0f * sigma * sigma);
    for (int k = 0; k < n_sym; k++) {
        float met[64];
        for (int s = 0; s < P; s++) {
            float dr = sym[2 * k] - cstl[2 * s], di = sym[2 * k + 1] - cstl[2 * s + 1];
            met[s] = -(dr * dr + di * di) * inv;
        }
        for (int b = 0; b < bps; b++) {
            float L0 = -INFINITY, L1 = -INFINITY;
            for (int s = 0; s < P; s++)
                if (((s >> b) & 1) == 0) L0 = max_star(L0, met[s]); else L1 = max_star(L1, met[s]);
            llr[k * bps + b] = L0 - L1;
        }
    }
}

void orc_itl_lut(int N, int n_cols, int order, uint32_t *lut)
{
    if (n_cols <= 1) { for (int i = 0; i < N; i++) lut[i] = (uint32_t)i; return; }
    int n_rows = N / n_cols;
    for (int i = 0; i < n_rows; i++)
        for (int j = 0; j < n_cols; j++)
            lut[i * n_cols + j] = (uint32_t)((order == 0 ? j : (n_cols - 1 - j)) * n_rows + i);
}

/* ======================================================================== glue */
void orc_pl_rand_seq(int n, uint8_t *seq)
{
    /* ETSI EN 302 307 5.5.4: x(i+18)=x(i+7)+x(i), y(i+18)=y(i+10)+y(i+7)+y(i+5)+y(i);
     * z_n(i) = x((i+n) mod (2^18-1)) + y(i); R_n(i) = 2 z_n((i+131072) mod (2^18-1)) + z_n(i) */
    const int P = (1 << 18) - 1;
    uint8_t *x = (uint8_t *)malloc(P), *y = (uint8_t *)malloc(P);
    for (int i = 0; i < 18; i++) { x[i] = (i == 0); y[i] = 1; }
    for (int i = 0; i + 18 < P; i++) {
        x[i + 18] = x[i + 7] ^ x[i];
        y[i + 18] = y[i + 10] ^ y[i + 7] ^ y[i + 5] ^ y[i];
    }
    for (int i = 0; i < 66420; i++) {
        int z0 = x[(i + n) % P] ^ y[i];
        int i2 = (i + 131072) % P;
        int z1 = x[(i2 + n) % P] ^ y[i2];
        seq[i] = (uint8_t)(2 * z1 + z0);
    }
    free(x); free(y);
}

void orc_pl_scramble(const float *in, float *out, int n_sym, int start_ix, int scramble)
{
    static uint8_t *seq = NULL;
    if (!seq) { seq = (uint8_t *)malloc(66420); orc_pl_rand_seq(0, seq); }
    for (int i = 0; i < 2 * start_ix; i++) out[i] = in[i];
    for (int i = start_ix; i < n_sym; i++) {
        int R = seq[i - start_ix];
        int lsb = R % 2, msb = R / 2;
        int rr = (1 - lsb) * (-2 * msb + 1);
        int ri = lsb * (-2 * msb + 1);
        ri = (2 * (scramble ? 1 : 0) - 1) * ri;
        float dr = in[2 * i], di = in[2 * i + 1];
        out[2 * i]     = rr * dr - ri * di;
        out[2 * i + 1] = ri * dr + rr * di;
    }
}

void orc_bb_scramble(const int32_t *in, int32_t *out, int n)
{
    /* 1 + x^14 + x^15, init 100101010000000, restarted every frame */
    int lfsr[15] = {1, 0, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        int fb = (lfsr[14] + lfsr[13]) % 2;
        for (int j = 14; j > 0; j--) lfsr[j] = lfsr[j - 1];
        lfsr[0] = fb;
        out[i] = (in[i] + fb) % 2;
    }
}

void orc_plheader(const int *mod_cod, float *plh)
{
    static const int G[7][32] = {
        {1,0,0,1,0,0,0,0,1,0,1,0,1,1,0,0,0,0,1,0,1,1,0,1,1,1,0,1,1,1,0,1},
        {0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1},
        {0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1},
        {0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,1,1,1,1},
        {0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1},
        {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1},
        {1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1}};
    static const int PLS_SCR[64] = {0,1,1,1,0,0,0,1,1,0,0,1,1,1,0,1,1,0,0,0,0,0,1,1,1,1,0,0,1,0,0,1,
                                    0,1,0,1,0,0,1,1,0,1,0,0,0,0,1,0,0,0,1,0,1,1,0,1,1,1,1,1,1,0,1,0};
    static const int SOF[26] = {0,1,1,0,0,0,1,1,0,1,0,0,1,0,1,1,1,0,1,0,0,0,0,0,1,0};
    const float a = (float)(1 / sqrt(2.0));
    for (int i = 0; i < 13; i++) {
        int e = 1 - 2 * SOF[2 * i], o = 1 - 2 * SOF[2 * i + 1];
        plh[4 * i] = a * e; plh[4 * i + 1] = a * e; plh[4 * i + 2] = -1 * a * o; plh[4 * i + 3] = a * o;
    }
    int coded[32], fin[64];
    for (int c = 0; c < 32; c++) { int s = 0; for (int r = 0; r < 7; r++) s = (s + mod_cod[r] * G[r][c]) % 2; coded[c] = s; }
    for (int i = 0; i < 32; i++) {
        fin[2 * i]     = (coded[i] + PLS_SCR[2 * i]) % 2;
        fin[2 * i + 1] = ((coded[i] == 0 ? 1 : 0) + PLS_SCR[2 * i + 1]) % 2;
    }
    float *p = plh + 52;
    for (int i = 0; i < 32; i++) {
        int e = 1 - 2 * fin[2 * i], o = 1 - 2 * fin[2 * i + 1];
        if (mod_cod[0] == 0) { p[4*i] = a*e; p[4*i+1] = a*e; p[4*i+2] = -1*a*o; p[4*i+3] = a*o; }
        else                 { p[4*i] = -1*a*e; p[4*i+1] = a*e; p[4*i+2] = -1*a*o; p[4*i+3] = -1*a*o; }
    }
}

int orc_pl_frame_size(int n_xfec_sym)
{
    const int M = 90, P = 36;
    int S = n_xfec_sym / M, np = n_xfec_sym / (16 * M);
    return M * (S + 1) + np * P;                       /* DVBS2.cpp:351-355 */
}

void orc_framer_generate(const float *xfec, int n_xfec_sym, const float *plh, float *plf)
{
    const int M = 90, P = 36; int np = n_xfec_sym / (16 * M);
    const float a = (float)(1 / sqrt(2.0));
    int o = 0;
    for (int i = 0; i < 180; i++) plf[o++] = plh[i];
    int d = 0;
    for (int b = 0; b < np; b++) {
        for (int i = 0; i < 2 * 16 * M; i++) plf[o++] = xfec[d++];
        for (int i = 0; i < P; i++) { plf[o++] = a; plf[o++] = a; }
    }
    while (d < 2 * n_xfec_sym) plf[o++] = xfec[d++];
}

void orc_framer_remove_plh(const float *plf, int n_xfec_sym, float *xfec)
{
    const int M = 90, P = 36; int np = n_xfec_sym / (16 * M);
    int s = 2 * M, d = 0;
    for (int b = 0; b < np; b++) {
        for (int i = 0; i < 2 * 16 * M; i++) xfec[d++] = plf[s++];
        s += 2 * P;
    }
    while (d < 2 * n_xfec_sym) xfec[d++] = plf[s++];
}

void orc_estimate(const float *x, int n_sym, float code_rate, int bps, float *out3)
{
    float m2 = 0, m4 = 0;
    for (int i = 0; i < n_sym; i++) {
        float tmp = x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1];
        m2 += tmp; m4 += tmp * tmp;
    }
    m2 /= n_sym; m4 /= n_sym;
    float Se = sqrtf(fabsf(2 * m2 * m2 - m4));
    float Ne = fabsf(m2 - Se);
    float esn0 = 10 * log10f(Se / Ne);
    if (isinf(esn0)) esn0 = 100.f;
    /* tools::esn0_to_sigma (upsample 1) and esn0_to_ebn0 */
    float sigma = sqrtf(1.0f / (2.0f * powf(10.0f, esn0 / 10.0f)));
    float ebn0 = esn0 - 10.0f * log10f(code_rate * (float)bps);
    out3[0] = sigma; out3[1] = ebn0; out3[2] = esn0;
}

void orc_agc(const float *x, int n_cplx, float output_energy, float *z)
{
    float sum_abs_2 = 0.0f, sum_re = 0.0f, sum_im = 0.0f;
    for (int i = 0; i < n_cplx; i++) {
        sum_abs_2 += x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1];
        sum_re += x[2 * i];
        sum_im += x[2 * i + 1];
    }
    float std_xn = sqrtf(sum_abs_2 * (float)n_cplx - sum_re * sum_re - sum_im * sum_im) / (float)n_cplx;
    std_xn /= sqrtf(output_energy);
    for (int i = 0; i < n_cplx; i++) { z[2 * i] = x[2 * i] / std_xn; z[2 * i + 1] = x[2 * i + 1] / std_xn; }
}

void orc_nco(const float *x, int n_cplx, float nu, float *n, float *z)
{
    const float new_nu = floorf(nu * 1e6f) / 1e6f;
    const float omega = (float)(2 * 3.1415926535897932384626433832795 * new_nu);
    for (int i = 0; i < n_cplx; i++) {
        const float phase = omega * *n, c = cosf(phase), s = sinf(phase);
        z[2 * i] = x[2 * i] * c - x[2 * i + 1] * s;
        z[2 * i + 1] = x[2 * i] * s + x[2 * i + 1] * c;
        *n = (*n >= 999999.f) ? 0.f : *n + 1.f;
    }
}

void orc_rrc_taps(float rolloff, int osf, int grp, float *taps)
{
    const float PI = (float)3.1415926535897932384626433832795;
    int c = grp * osf, T = 2 * c + 1;
    float eps = FLT_EPSILON;
    taps[c] = 1.0f - rolloff + 4.0f * rolloff / PI;
    float en = taps[c] * taps[c];
    for (int i = 1; i <= c; i++) {
        float t = (float)i / (float)osf, v;
        if (fabsf(4.0f * rolloff * t - 1.0f) <= eps || fabsf(4.0f * rolloff * t + 1.0f) <= eps)
            v = rolloff / sqrtf(2.0f) * ((1.0f + 2.0f / PI) * sinf(PI / (4.0f * rolloff)) +
                                         (1.0f - 2.0f / PI) * cosf(PI / (4.0f * rolloff)));
        else {
            float den = PI * t * (1.0f - 16.0f * rolloff * rolloff * t * t);
            float num = sinf(PI * t * (1.0f - rolloff)) + 4.0f * rolloff * t * cosf(PI * t * (1.0f + rolloff));
            v = num / den;
        }
        taps[c + i] = v; taps[c - i] = v;
        en += v * v + v * v;
    }
    for (int i = 0; i < T; i++) taps[i] /= sqrtf(en);
}

void orc_fir(const float *taps, int T, float *hist, const float *x, float *y, int n)
{
    /* y[i] = sum_k brev[k] x[i-(T-1)+k], brev[k] = taps[T-1-k] (Filter_FIR_ccr.cpp:26-27);
     * x[<0] = tail of the previous frame */
    const int H = T - 1;
    float *ext = (float *)malloc(sizeof(float) * 2 * (n + H));
    memcpy(ext, hist, sizeof(float) * 2 * H);
    memcpy(ext + 2 * H, x, sizeof(float) * 2 * n);
    for (int i = 0; i < n; i++) {
        float ar = ext[2 * i] * taps[T - 1], ai = ext[2 * i + 1] * taps[T - 1];
        for (int k = 1; k < T; k++) {
            ar += ext[2 * (i + k)] * taps[T - 1 - k];
            ai += ext[2 * (i + k) + 1] * taps[T - 1 - k];
        }
        y[2 * i] = ar; y[2 * i + 1] = ai;
    }
    memcpy(hist, ext + 2 * n, sizeof(float) * 2 * H);
    free(ext);
}

void orc_upfir(const float *taps, int T, int osf, float *hist, const float *x, float *y, int n_in)
{
    float *up = (float *)calloc((size_t)2 * n_in * osf, sizeof(float));
    for (int i = 0; i < n_in; i++) { up[2 * i * osf] = x[2 * i]; up[2 * i * osf + 1] = x[2 * i + 1]; }
    orc_fir(taps, T, hist, up, y, n_in * osf);
    free(up);
}

/* ================================================================ fine frequency / phase synchronizers (row N4) */
#ifndef M_PI
#define M_PI 3.1415926535897932384626433832795
#endif
int orc_sff_pilots(int n_cplx, int *pilot_start, int max)
{
    int n = 0;
    for (int idx = 1530; idx < n_cplx; idx += 1476) { if (n < max) pilot_start[n] = idx; n++; }   /* .cpp:18-24 */
    return n;
}

float orc_lr_synchronize(int n_cplx, float alpha, float *R_l, const float *X, float *Y)
{
    int ps[64];
    const int P = orc_sff_pilots(n_cplx, ps, 64), Lp = 36 / 2, Lp_2 = Lp / 2;
    float t0 = 0.f, t1 = 0.f;
    for (int p = 0; p < P; p++) {                                                   /* .cpp:102-130 */
        float z[2 * 18];
        for (int i = 0; i < Lp; i++) {
            z[2 * i] = X[2 * i + 2 * ps[p]] + X[2 * i + 2 * ps[p] + 1];
            z[2 * i + 1] = X[2 * i + 2 * ps[p] + 1] - X[2 * i + 2 * ps[p]];
        }
        for (int m = 1; m < Lp_2 + 1; m++) {
            float s0 = 0.f, s1 = 0.f;
            for (int k = m; k < Lp; k++) {
                s0 += z[2 * k] * z[2 * (k - m)] + z[2 * k + 1] * z[2 * (k - m) + 1];
                s1 += z[2 * k + 1] * z[2 * (k - m)] - z[2 * k] * z[2 * (k - m) + 1];
            }
            t0 += s0 / (float)(2 * (Lp - m));
            t1 += s1 / (float)(2 * (Lp - m));
        }
    }
    R_l[0] = alpha * R_l[0] + (1 - alpha) * t0;                                     /* .cpp:131-132 */
    R_l[1] = alpha * R_l[1] + (1 - alpha) * t1;
    float est = atan2f(R_l[1], R_l[0]);                                             /* .cpp:133 */
    est = (float)(est / ((Lp_2 + 1) * M_PI));                                       /* .cpp:134 */
    const float f_ = (float)(est * M_PI);                                           /* .cpp:135 */
    for (int n = 0; n < 2 * n_cplx; n += 2) {                                       /* .cpp:141-166 (mipp::cossin / std::cos, std::sin) */
        const float theta = f_ * (float)n, c = cosf(theta), sn = sinf(theta);
        Y[n] = X[n] * c + X[n + 1] * sn;
        Y[n + 1] = X[n + 1] * c - X[n] * sn;
    }
    return est;
}

void orc_fp_synchronize(int n_cplx, const float *X, float *Y, float *out2)
{
    int ps[64];
    const int P = orc_sff_pilots(n_cplx, ps, 64), Lp = 36;
    const float inv_2PI = (float)(1.0f / (2 * M_PI));                               /* .cpp:50 */
    float phase_est[64] = {0}, y[64], t[64];
    if (P < 1 || P > 64) { memcpy(Y, X, sizeof(float) * 2 * (size_t)n_cplx); out2[0] = out2[1] = 0.f; return; }   /* no pilot block: the reference would divide by zero */
    for (int p = 0; p < P; p++) {                                                   /* .cpp:53-67 */
        float s0 = 0.f, s1 = 0.f;
        for (int i = 0; i < Lp; i++) {
            s0 += X[2 * ps[p] + 2 * i] + X[2 * ps[p] + 2 * i + 1];
            s1 += X[2 * ps[p] + 2 * i + 1] - X[2 * ps[p] + 2 * i];
        }
        phase_est[p] = atan2f(s1, s0);
        phase_est[p] = phase_est[p] < 0 ? (float)(phase_est[p] + 2 * M_PI) : phase_est[p];
    }
    y[0] = inv_2PI * phase_est[0];                                                   /* .cpp:72-73 */
    t[0] = ps[0] + (float)(Lp / 2);
    float acc = 0.f;
    for (int p = 1; p < P; p++) {                                                    /* .cpp:77-85 */
        const float diff_angle = phase_est[p] - phase_est[p - 1];
        float acc_elt = diff_angle > 0 ? floorf(diff_angle * inv_2PI + 0.5f) : ceilf(diff_angle * inv_2PI - 0.5f);
        acc_elt = fabsf(diff_angle) > M_PI ? acc_elt : 0.0f;
        acc += acc_elt;
        y[p] = inv_2PI * phase_est[p] - acc;
        t[p] = ps[p] + (float)(Lp / 2);
    }
    float sum_t = 0.f, sum_y = 0.f, sum_ty = 0.f, sum_tt = 0.f;
    for (int p = 0; p < P; p++) { sum_t += t[p]; sum_y += y[p]; sum_ty += t[p] * y[p]; sum_tt += t[p] * t[p]; }    /* .cpp:92-98 */
    const float ef = (P * sum_ty - sum_t * sum_y) / (P * sum_tt - sum_t * sum_t);   /* .cpp:100 */
    const float ep = (sum_y - ef * sum_t) / P;                                      /* .cpp:101 */
    out2[0] = ef; out2[1] = ep;
    for (int n = 0; n < n_cplx; n++) {                                              /* .cpp:103-112 */
        const float theta = (float)(2 * M_PI * (ef * (float)n + ep));
        Y[2 * n] = X[2 * n] * cosf(theta) + X[2 * n + 1] * sinf(theta);
        Y[2 * n + 1] = X[2 * n + 1] * cosf(theta) - X[2 * n] * sinf(theta);
    }
}

/* ================================================================ frame synchronizer (row N4) */
/* Variable_delay_cc_naive: ctor Variable_delay_cc_naive.cpp:11-24, `_filter` :56-79 */
struct orc_vdelay { int N, delay, size, head2, first_time, nbuff2; float *buff2; };

orc_vdelay *orc_vdelay_create(int N, int delay, int max_delay)
{
    orc_vdelay *d = (orc_vdelay *)calloc(1, sizeof *d);
    d->N = N; d->delay = delay; d->size = max_delay + 1; d->head2 = 0; d->first_time = 1;
    d->nbuff2 = 4 * (max_delay + 1);
    d->buff2 = (float *)calloc((size_t)d->nbuff2, sizeof(float));
    return d;
}
void orc_vdelay_destroy(orc_vdelay *d) { if (d) { free(d->buff2); free(d); } }
void orc_vdelay_set_delay(orc_vdelay *d, int delay) { d->delay = delay < d->size ? delay : d->size - 1; }   /* .cpp:91-95 */
void orc_vdelay_reset(orc_vdelay *d)       /* Variable_delay_cc_naive.cpp reset(): buffers to zero, heads to 0 */
{
    memset(d->buff2, 0, sizeof(float) * (size_t)d->nbuff2);
    d->head2 = 0; d->first_time = 1;
}
void orc_vdelay_filter(orc_vdelay *d, const float *X, float *Y)
{
    const int N = d->N, D = 2 * d->delay;
    int start_Y = (D > d->head2) ? D - d->head2 : 0;
    int start_buff = (D < d->head2) ? d->head2 - D : 0;
    int end_buff = start_buff + D;
    end_buff = end_buff > d->nbuff2 ? d->nbuff2 : end_buff;
    end_buff = (end_buff - start_buff > N - start_Y) ? end_buff - ((end_buff - start_buff) - (N - start_Y)) : end_buff;
    if (start_Y && !d->first_time) memmove(Y, Y + N - start_Y, sizeof(float) * (size_t)start_Y);
    else memset(Y, 0, sizeof(float) * (size_t)start_Y);
    memcpy(Y + start_Y, d->buff2 + start_buff, sizeof(float) * (size_t)(end_buff - start_buff));
    memcpy(Y + D, X, sizeof(float) * (size_t)(N - D));
    memcpy(d->buff2, X + N - D, sizeof(float) * (size_t)D);
    d->first_time = 0;
    d->head2 = D;
}

static const float ORC_CONJ_SOF[25] = {1, -1, -1, 1, -1, 1, 1, -1, 1, 1, -1, -1, 1, -1, -1, -1, 1, -1, -1, -1, -1, 1, 1, 1, 1};
static const float ORC_CONJ_PLSC[64] = {1, 0, 1, 0, -1, 0, -1, 0, 1, 0, -1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1, 0, 1, 0, -1, 0, -1, 0, 1, 0, 1, 0,
                                        1, 0, 1, 0, -1, 0, -1, 0, -1, 0, -1, 0, -1, 0, 1, 0, 1, 0, -1, 0, 1, 0, 1, 0, 1, 0, -1, 0, -1, 0, 1, 0};
void orc_sfm_taps(const float **sof, int *n_sof, const float **plsc, int *n_plsc)
{
    *sof = ORC_CONJ_SOF; *n_sof = 25; *plsc = ORC_CONJ_PLSC; *n_plsc = 64;
}

struct orc_sfm {
    int n;                       /* complex samples per frame */
    float alpha, trigger, max_corr;
    int vecw, delay;
    float reg_re, reg_im;        /* reg_channel */
    float *corr_vec;             /* [n] */
    float *hist_sof, *hist_plsc; /* FIR memories of corr_SOF (24) / corr_PLSC (63) */
    orc_vdelay *output_delay, *sof_plsc_delay;
    float *diff, *cor_sof, *cor_sof_delayed, *cor_plsc;
};

orc_sfm *orc_sfm_create(int n_cplx, float alpha, float trigger, int vec_width)
{
    orc_sfm *s = (orc_sfm *)calloc(1, sizeof *s);
    const int N = 2 * n_cplx;
    s->n = n_cplx; s->alpha = alpha; s->trigger = trigger; s->vecw = vec_width > 0 ? vec_width : 1;
    s->reg_re = 1.f; s->reg_im = 0.f;                                             /* .cpp:19 */
    s->corr_vec = (float *)calloc((size_t)n_cplx, sizeof(float));                 /* .cpp:20 */
    s->hist_sof = (float *)calloc(2 * 24, sizeof(float)); s->hist_plsc = (float *)calloc(2 * 63, sizeof(float));
    s->output_delay = orc_vdelay_create(N, N / 2, N / 2);                         /* .cpp:21 */
    s->sof_plsc_delay = orc_vdelay_create(N, 64, 64);                             /* .cpp:24 */
    s->diff = (float *)calloc((size_t)N, sizeof(float)); s->cor_sof = (float *)calloc((size_t)N, sizeof(float));
    s->cor_sof_delayed = (float *)calloc((size_t)N, sizeof(float)); s->cor_plsc = (float *)calloc((size_t)N, sizeof(float));
    return s;
}
void orc_sfm_destroy(orc_sfm *s)
{
    if (!s) return;
    free(s->corr_vec); free(s->hist_sof); free(s->hist_plsc); orc_vdelay_destroy(s->output_delay); orc_vdelay_destroy(s->sof_plsc_delay);
    free(s->diff); free(s->cor_sof); free(s->cor_sof_delayed); free(s->cor_plsc); free(s);
}
void orc_sfm_reset(orc_sfm *s)        /* .cpp:304-318 (SOF_PLSC_delay is not reset there) */
{
    orc_vdelay_reset(s->output_delay); orc_vdelay_set_delay(s->output_delay, 0);
    s->reg_re = 1.f; s->reg_im = 0.f;
    memset(s->corr_vec, 0, sizeof(float) * (size_t)s->n);
    memset(s->hist_sof, 0, sizeof(float) * 2 * 24); memset(s->hist_plsc, 0, sizeof(float) * 2 * 63);
}
float orc_sfm_metric(const orc_sfm *s) { return s->max_corr; }
int orc_sfm_packet_flag(const orc_sfm *s) { return s->max_corr > s->trigger; }

void orc_sfm_synchronize1(orc_sfm *s, const float *X, float *cor_SOF, float *cor_PLSC)
{
    const int n = s->n;
    float *d = s->diff;
    d[0] = s->reg_re * X[0] + s->reg_im * X[1];                                   /* .cpp:136-137 */
    d[1] = s->reg_im * X[0] - s->reg_re * X[1];
    for (int i = 1; i < n; i++) {                                                 /* .cpp:138-142 */
        d[2 * i] = X[2 * i - 2] * X[2 * i] + X[2 * i - 1] * X[2 * i + 1];
        d[2 * i + 1] = X[2 * i - 1] * X[2 * i] - X[2 * i - 2] * X[2 * i + 1];
    }
    orc_fir(ORC_CONJ_SOF, 25, s->hist_sof, d, cor_SOF, n);                        /* .cpp:144 */
    orc_fir(ORC_CONJ_PLSC, 64, s->hist_plsc, d, cor_PLSC, n);                     /* .cpp:145 */
}

int orc_sfm_synchronize2(orc_sfm *s, const float *X, const float *cor_SOF, const float *cor_PLSC, float *Y)
{
    const int n = s->n;
    orc_vdelay_filter(s->sof_plsc_delay, cor_SOF, s->cor_sof_delayed);            /* .cpp:236 */
    float max_corr = 0.f;
    int max_idx = 0;
    const int end_vec = (n / s->vecw) * s->vecw;
    const float *cs = s->cor_sof_delayed;
    for (int i = 0; i < n; i++) {
        const float sr = cor_PLSC[2 * i] + cs[2 * i], si = cor_PLSC[2 * i + 1] + cs[2 * i + 1];
        const float dr = cs[2 * i] - cor_PLSC[2 * i], di = cs[2 * i + 1] - cor_PLSC[2 * i + 1];
        const float a2s = fmaf(sr, sr, si * si), a2d = fmaf(dr, dr, di * di);      /* mipp::fmadd, .cpp:253-261 */
        const float m = sqrtf(a2s > a2d ? a2s : a2d);
        if (i < end_vec) s->corr_vec[i] = s->alpha * s->corr_vec[i] + (1.f - s->alpha) * m;   /* .cpp:263-267 */
        else s->corr_vec[i] = m;                                                  /* the second assignment wins, .cpp:284-285 */
        if (s->corr_vec[i] > max_corr) { max_corr = s->corr_vec[i]; max_idx = i; } /* .cpp:269-276, :287-291 */
    }
    s->max_corr = max_corr;
    s->reg_re = X[2 * n - 2]; s->reg_im = X[2 * n - 1];                            /* .cpp:294 */
    const int delay = (n + max_idx - 25 - 64) % n;                                /* .cpp:296 */
    s->delay = delay;
    orc_vdelay_set_delay(s->output_delay, (n - delay) % n);                       /* .cpp:298 */
    orc_vdelay_filter(s->output_delay, X, Y);                                     /* .cpp:299 */
    return delay;
}

int orc_sfm_synchronize(orc_sfm *s, const float *X, float *Y)                     /* .cpp:46-128: the same two steps in one task */
{
    orc_sfm_synchronize1(s, X, s->cor_sof, s->cor_plsc);
    return orc_sfm_synchronize2(s, X, s->cor_sof, s->cor_plsc, Y);
}
