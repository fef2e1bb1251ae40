// The two functions of `--dec-implem SPA_TANH` (AFF3CT's tanh-product check node), shared by k_ldpc_wg8.hip (QC layers) and k_ldpc_nat.hip (natural row order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dvbs2 {

// `--dec-implem SPA_TANH`: the check node in the form AFF3CT's Update_rule_SPA evaluates [UPSTREAM-RECALL, oracle/dvbs2_oracle.c chk_update_spa_tanh]:
//     t_j = tanh(|v_j| / 2) in fp32 (1.0f beyond 18.02),  P = prod t_j in the oracle's edge order,  val = P / t_j clamped to 1 - 2^-23,  |out_j| = 2 atanh(val).
// Near the cap the quotient moves in steps of 2^-24 = the message in steps of 0.1 .. 0.7: a twin of the oracle has to agree bit for bit.  tanh, the quotients and
// log1p are therefore made of operations whose IEEE-754 result is correctly rounded (add, multiply, fma, divide, v_rndne) in exactly the oracle's order -- no
// v_exp / v_log / v_rcp here (the build has no fast-math flag and -ffp-contract=off: `/` is the correctly rounded division sequence).
__device__ __forceinline__ float w8_det_expm1(float y)            // e^y - 1, -2.1 <= y <= 45
{
    const float n = __builtin_rintf(y * 1.44269502f);
    float r = __builtin_fmaf(-n, 0.693145751953125f, y);
    r = __builtin_fmaf(-n, 1.42860677e-6f, r);
    float q = 1.98412701e-4f;
    q = __builtin_fmaf(q, r, 1.38888892e-3f);
    q = __builtin_fmaf(q, r, 8.33333377e-3f);
    q = __builtin_fmaf(q, r, 4.16666679e-2f);
    q = __builtin_fmaf(q, r, 1.66666672e-1f);
    q = __builtin_fmaf(q, r, 0.5f);
    const float pm1 = __builtin_fmaf(q * r, r, r);
    const float sc = __uint_as_float((uint32_t)((int)n + 127) << 23);
    return __builtin_fmaf(sc, pm1, sc - 1.0f);
}
__device__ __forceinline__ float w8_det_tanh_half(float a)        // tanh(a / 2), a >= 0 (+inf: absent / NULL slots -> exactly 1); one expm1 and one division on either branch of the oracle's
{
    const bool big = a >= 2.0f;
    const float ac = fminf(a, 44.0f);
    const float t = w8_det_expm1(big ? ac : -ac);
    const float d = (big ? 2.0f : -t) / (t + 2.0f);
    const float r = big ? 1.0f - d : d;
    return (a < 44.0f) ? r : 1.0f;
}
__device__ __forceinline__ float w8_det_log1p(float w)            // log(1 + w), 0 <= w < 2^26
{
    const float u = 1.0f + w;
    const float c = w - (u - 1.0f);
    const uint32_t iu = __float_as_uint(u);
    int e = (int)(iu >> 23) - 127;
    uint32_t im = (iu & 0x007FFFFFu) | 0x3F800000u;
    const bool up = im >= 0x3FB504F3u;
    im = up ? im - 0x00800000u : im;
    e = up ? e + 1 : e;
    const float f = __builtin_fmaf(c, __uint_as_float((uint32_t)(127 - e) << 23), __uint_as_float(im) - 1.0f);
    const float s = f / (2.0f + f);
    const float z = s * s;
    float q = 0.111111112f;
    q = __builtin_fmaf(q, z, 0.142857149f);
    q = __builtin_fmaf(q, z, 0.2f);
    q = __builtin_fmaf(q, z, 0.333333343f);
    const float s2 = s + s;
    const float lm = __builtin_fmaf(s2 * z, q, s2);
    const float fe = (float)e;
    float r = __builtin_fmaf(fe, 0.693145751953125f, lm);
    r = __builtin_fmaf(fe, 1.42860677e-6f, r);
    return r;
}
}  // namespace dvbs2
