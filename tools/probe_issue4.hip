// Round 4, fourth issue probe: what do the OTHER 64-bit encodings cost -- SDWA (a VOP1 / VOP2 / VOPC opcode with an |abs| modifier WITHOUT the VOP3 form) and DPP --
// and the carry / vcc readers of the 32-bit forms?  Same harness as probe_issue2: eight independent registers, W = 1 .. 4 waves on one SIMD, time of every wave.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_issue4.hip -o tools/bin/probe_issue4 && timeout 120 tools/bin/probe_issue4
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define EIGHT(pre, post) pre "%0" post "\n" pre "%1" post "\n" pre "%2" post "\n" pre "%3" post "\n" pre "%4" post "\n" pre "%5" post "\n" pre "%6" post "\n" pre "%7" post "\n"
#define SD " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, float *sink, int iters, float seed)
{
    float r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, c = seed * 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define RUN(txt) asm volatile(REP16(txt) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c) : "vcc")
    for (int i = 0; i < iters; i++) {
        if (OP == 0) RUN("v_min_f32_sdwa %0, |%0|, %8" SD "\n v_min_f32_sdwa %1, |%1|, %8" SD "\n v_min_f32_sdwa %2, |%2|, %8" SD "\n v_min_f32_sdwa %3, |%3|, %8" SD "\n v_min_f32_sdwa %4, |%4|, %8" SD "\n v_min_f32_sdwa %5, |%5|, %8" SD "\n v_min_f32_sdwa %6, |%6|, %8" SD "\n v_min_f32_sdwa %7, |%7|, %8" SD "\n");
        if (OP == 1) RUN("v_cmp_eq_f32_sdwa vcc, |%0|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%1|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%2|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%3|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%4|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%5|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%6|, %8 src0_sel:DWORD src1_sel:DWORD\n v_cmp_eq_f32_sdwa vcc, |%7|, %8 src0_sel:DWORD src1_sel:DWORD\n");
        if (OP == 2) RUN(EIGHT("v_cndmask_b32_e32 ", ", %8, %8, vcc") "s_nop 0\n" );
        if (OP == 3) RUN("v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %4, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %5, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %6, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %7, vcc\n");
        if (OP == 4) RUN("v_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n");
        if (OP == 5) RUN("v_fmac_f32_e32 %0, %8, %8\n v_fmac_f32_e32 %1, %8, %8\n v_fmac_f32_e32 %2, %8, %8\n v_fmac_f32_e32 %3, %8, %8\n v_fmac_f32_e32 %4, %8, %8\n v_fmac_f32_e32 %5, %8, %8\n v_fmac_f32_e32 %6, %8, %8\n v_fmac_f32_e32 %7, %8, %8\n");
        if (OP == 6) RUN("v_cmp_eq_f32_e32 vcc, %0, %8\n v_cmp_eq_f32_e32 vcc, %1, %8\n v_cmp_eq_f32_e32 vcc, %2, %8\n v_cmp_eq_f32_e32 vcc, %3, %8\n v_cmp_eq_f32_e32 vcc, %4, %8\n v_cmp_eq_f32_e32 vcc, %5, %8\n v_cmp_eq_f32_e32 vcc, %6, %8\n v_cmp_eq_f32_e32 vcc, %7, %8\n");
        if (OP == 7) RUN("v_cmp_eq_f32_e32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %8, %1, vcc\n v_cmp_eq_f32_e32 vcc, %2, %8\n v_cndmask_b32_e32 %3, %8, %3, vcc\n v_cmp_eq_f32_e32 vcc, %4, %8\n v_cndmask_b32_e32 %5, %8, %5, vcc\n v_cmp_eq_f32_e32 vcc, %6, %8\n v_cndmask_b32_e32 %7, %8, %7, vcc\n");
        if (OP == 8) RUN("v_and_b32_sdwa %0, %0, %8" SD "\n v_and_b32_sdwa %1, %1, %8" SD "\n v_and_b32_sdwa %2, %2, %8" SD "\n v_and_b32_sdwa %3, %3, %8" SD "\n v_and_b32_sdwa %4, %4, %8" SD "\n v_and_b32_sdwa %5, %5, %8" SD "\n v_and_b32_sdwa %6, %6, %8" SD "\n v_and_b32_sdwa %7, %7, %8" SD "\n");
        if (OP == 9) RUN("v_max_f32_e64 %0, |%0|, |%0|\n v_max_f32_e64 %1, |%1|, |%1|\n v_max_f32_e64 %2, |%2|, |%2|\n v_max_f32_e64 %3, |%3|, |%3|\n v_max_f32_e64 %4, |%4|, |%4|\n v_max_f32_e64 %5, |%5|, |%5|\n v_max_f32_e64 %6, |%6|, |%6|\n v_max_f32_e64 %7, |%7|, |%7|\n");
        if (OP == 10) RUN("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n");
        if (OP == 11) { RUN("v_lshlrev_b32_e32 %0, 1, %0\n v_lshlrev_b32_e32 %1, 1, %1\n v_lshlrev_b32_e32 %2, 1, %2\n v_lshlrev_b32_e32 %3, 1, %3\n v_lshlrev_b32_e32 %4, 1, %4\n v_lshlrev_b32_e32 %5, 1, %5\n v_lshlrev_b32_e32 %6, 1, %6\n v_lshlrev_b32_e32 %7, 1, %7\n "); }
        if (OP == 12) { RUN("v_mov_b32_e32 %0, %8\n v_mov_b32_e32 %1, %8\n v_mov_b32_e32 %2, %8\n v_mov_b32_e32 %3, %8\n v_mov_b32_e32 %4, %8\n v_mov_b32_e32 %5, %8\n v_mov_b32_e32 %6, %8\n v_mov_b32_e32 %7, %8\n "); }
        if (OP == 13) { RUN("v_and_b32_e32 %0, %8, %0\n v_and_b32_e32 %1, %8, %1\n v_and_b32_e32 %2, %8, %2\n v_and_b32_e32 %3, %8, %3\n v_and_b32_e32 %4, %8, %4\n v_and_b32_e32 %5, %8, %5\n v_and_b32_e32 %6, %8, %6\n v_and_b32_e32 %7, %8, %7\n "); }
        if (OP == 14) { RUN("v_min_f32_e32 %0, %8, %0\n v_min_f32_e32 %1, %8, %1\n v_min_f32_e32 %2, %8, %2\n v_min_f32_e32 %3, %8, %3\n v_min_f32_e32 %4, %8, %4\n v_min_f32_e32 %5, %8, %5\n v_min_f32_e32 %6, %8, %6\n v_min_f32_e32 %7, %8, %7\n "); }
        if (OP == 15) { RUN("v_min_u32_e32 %0, %8, %0\n v_min_u32_e32 %1, %8, %1\n v_min_u32_e32 %2, %8, %2\n v_min_u32_e32 %3, %8, %3\n v_min_u32_e32 %4, %8, %4\n v_min_u32_e32 %5, %8, %5\n v_min_u32_e32 %6, %8, %6\n v_min_u32_e32 %7, %8, %7\n "); }
        if (OP == 16) { RUN("v_cvt_f32_i32_e32 %0, %0\n v_cvt_f32_i32_e32 %1, %1\n v_cvt_f32_i32_e32 %2, %2\n v_cvt_f32_i32_e32 %3, %3\n v_cvt_f32_i32_e32 %4, %4\n v_cvt_f32_i32_e32 %5, %5\n v_cvt_f32_i32_e32 %6, %6\n v_cvt_f32_i32_e32 %7, %7\n "); }
        if (OP == 17) { RUN("v_xor_b32_e32 %0, %8, %0\n v_xor_b32_e32 %1, %8, %1\n v_xor_b32_e32 %2, %8, %2\n v_xor_b32_e32 %3, %8, %3\n v_xor_b32_e32 %4, %8, %4\n v_xor_b32_e32 %5, %8, %5\n v_xor_b32_e32 %6, %8, %6\n v_xor_b32_e32 %7, %8, %7\n "); }
        if (OP == 18) { RUN("v_sub_f32_e32 %0, %8, %0\n v_sub_f32_e32 %1, %8, %1\n v_sub_f32_e32 %2, %8, %2\n v_sub_f32_e32 %3, %8, %3\n v_sub_f32_e32 %4, %8, %4\n v_sub_f32_e32 %5, %8, %5\n v_sub_f32_e32 %6, %8, %6\n v_sub_f32_e32 %7, %8, %7\n "); }
        if (OP == 19) { RUN("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n "); }
        if (OP == 20) { asm volatile("s_mov_b64 vcc, 0x55" ::: "vcc"); RUN("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n "); }
        if (OP == 21) { RUN("v_mul_u32_u24_e32 %0, %8, %0\n v_mul_u32_u24_e32 %1, %8, %1\n v_mul_u32_u24_e32 %2, %8, %2\n v_mul_u32_u24_e32 %3, %8, %3\n v_mul_u32_u24_e32 %4, %8, %4\n v_mul_u32_u24_e32 %5, %8, %5\n v_mul_u32_u24_e32 %6, %8, %6\n v_mul_u32_u24_e32 %7, %8, %7\n "); }
        if (OP == 22) { RUN("v_max_u32_e32 %0, %8, %0\n v_max_u32_e32 %1, %8, %1\n v_max_u32_e32 %2, %8, %2\n v_max_u32_e32 %3, %8, %3\n v_max_u32_e32 %4, %8, %4\n v_max_u32_e32 %5, %8, %5\n v_max_u32_e32 %6, %8, %6\n v_max_u32_e32 %7, %8, %7\n "); }
        if (OP == 23) { RUN("v_or_b32_e32 %0, %8, %0\n v_or_b32_e32 %1, %8, %1\n v_or_b32_e32 %2, %8, %2\n v_or_b32_e32 %3, %8, %3\n v_or_b32_e32 %4, %8, %4\n v_or_b32_e32 %5, %8, %5\n v_or_b32_e32 %6, %8, %6\n v_or_b32_e32 %7, %8, %7\n "); }
        if (OP == 24) { RUN("v_mul_f32_e32 %0, 2.0, %0\n v_mul_f32_e32 %1, 2.0, %1\n v_mul_f32_e32 %2, 2.0, %2\n v_mul_f32_e32 %3, 2.0, %3\n v_mul_f32_e32 %4, 2.0, %4\n v_mul_f32_e32 %5, 2.0, %5\n v_mul_f32_e32 %6, 2.0, %6\n v_mul_f32_e32 %7, 2.0, %7\n "); }
        if (OP == 25) { RUN("v_add_u32_e32 %0, 17, %0\n v_add_u32_e32 %1, 17, %1\n v_add_u32_e32 %2, 17, %2\n v_add_u32_e32 %3, 17, %3\n v_add_u32_e32 %4, 17, %4\n v_add_u32_e32 %5, 17, %5\n v_add_u32_e32 %6, 17, %6\n v_add_u32_e32 %7, 17, %7\n "); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    if (threadIdx.x % 64 == 0) { out[threadIdx.x / 64] = t1 - t0; out[16 + threadIdx.x / 64] = (hw >> 4) & 3u; }
    sink[threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}
template <int OP> static void run(const char *nm, unsigned long long *d, float *sink)
{
    const int iters = 400;
    for (int waves : {1, 2, 3, 4}) {
        const int threads = 256 * waves;
        probe<OP><<<1, threads>>>(d, sink, iters, 1.f); hipDeviceSynchronize();
        probe<OP><<<1, threads>>>(d, sink, iters, 1.f); hipDeviceSynchronize();
        unsigned long long h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-34s %d wave(s) per SIMD; cycles per instruction, waves of SIMD %llu:", nm, waves, h[16]);
        double mx = 0;
        for (int w = 0; w < 4 * waves; w++) if (h[16 + w] == h[16]) { printf(" %5.2f", (double)h[w] / (128.0 * iters)); if ((double)h[w] > mx) mx = (double)h[w]; }
        printf("   -> SIMD: one instruction per %.2f cycles\n", mx / (128.0 * iters) / waves);
        fflush(stdout);
    }
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long *d; float *sink; hipMalloc(&d, 32 * 8); hipMalloc(&sink, 1024 * 4);
    run<0>("v_min_f32_sdwa |abs|", d, sink);
    run<1>("v_cmp_eq_f32_sdwa vcc |abs|", d, sink);
    run<2>("v_cndmask_b32_e32 (vcc)", d, sink);
    run<3>("v_addc_co_u32_e32", d, sink);
    run<4>("v_add_f32_dpp quad_perm", d, sink);
    run<5>("v_fmac_f32_e32", d, sink);
    run<6>("v_cmp_eq_f32_e32", d, sink);
    run<7>("v_cmp_e32 + v_cndmask_e32 pairs", d, sink);
    run<8>("v_and_b32_sdwa (no modifier)", d, sink);
    run<9>("v_max_f32_e64 |abs| (VOP3)", d, sink);
    run<10>("v_mov_b32_dpp quad_perm", d, sink);
    run<11>("v_lshlrev_b32_e32", d, sink);
    run<12>("v_mov_b32_e32", d, sink);
    run<13>("v_and_b32_e32", d, sink);
    run<14>("v_min_f32_e32", d, sink);
    run<15>("v_min_u32_e32", d, sink);
    run<16>("v_cvt_f32_i32_e32", d, sink);
    run<17>("v_xor_b32_e32", d, sink);
    run<18>("v_sub_f32_e32", d, sink);
    run<19>("v_cndmask_b32_e32 dst=src0", d, sink);
    run<20>("v_cndmask_b32_e32 vcc set by s_mov", d, sink);
    run<21>("v_mul_u32_u24_e32", d, sink);
    run<22>("v_max_u32_e32", d, sink);
    run<23>("v_or_b32_e32", d, sink);
    run<24>("v_mul_f32 inline const", d, sink);
    run<25>("v_add_u32 inline const", d, sink);
    return 0;
}
