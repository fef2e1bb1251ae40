// Issue cost of the VALU instructions the LDPC layer loop is made of, measured per SIMD on gfx950: W waves per SIMD
// each run a block of 64 independent copies of one instruction 2000 times; cycles per instruction and SIMD =
// elapsed cycles / (W * 64 * 2000).  Decides which formulation of the min-sum update is cheapest
// (DESIGN.md section 4): hipcc --offload-arch=gfx950 -O2 tools/probe_valu.hip -o tools/bin/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, int iters, float seed)
{
    float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = 0.5f;
    unsigned u = OP >= 20 ? __builtin_amdgcn_readfirstlane(threadIdx.x) : threadIdx.x, v = 7u;
    unsigned long long m = 0x5555555555555555ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile(REP64("v_add_f32 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (OP == 1) asm volatile(REP64("v_cndmask_b32_e64 %0, %1, %2, %3\n") : "+v"(a) : "v"(b), "v"(c), "s"(m));
        if (OP == 2) asm volatile(REP64("v_cmp_eq_f32_e64 %0, %1, %2\n") : "+s"(m) : "v"(b), "v"(c));
        if (OP == 3) asm volatile(REP64("v_cmp_eq_f32_e32 vcc, %0, %1\n") : : "v"(b), "v"(c) : "vcc");
        if (OP == 4) asm volatile(REP64("v_cndmask_b32_e32 %0, %1, %2, vcc\n") : "+v"(a) : "v"(b), "v"(c) : "vcc");
        if (OP == 5) asm volatile(REP64("v_med3_f32 %0, %1, %2, %3\n") : "+v"(a) : "v"(b), "v"(c), "v"(d));
        if (OP == 6) asm volatile(REP64("v_min_f32 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (OP == 7) asm volatile(REP64("v_alignbit_b32 %0, %1, %2, 31\n") : "+v"(u) : "v"(v), "v"(b));
        if (OP == 8) asm volatile(REP64("v_and_or_b32 %0, %1, %2, %3\n") : "+v"(u) : "v"(v), "s"((unsigned)m), "v"(b));
        if (OP == 9) asm volatile(REP64("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c\n") : "+v"(u) : "s"((unsigned)m), "v"(v), "v"(b));
        if (OP == 10) asm volatile(REP64("v_lshlrev_b32 %0, 5, %1\n") : "+v"(u) : "v"(v));
        if (OP == 11) asm volatile(REP64("v_min_u32 %0, %1, %2\n") : "+v"(u) : "v"(v), "v"(b));
        if (OP == 12) asm volatile(REP64("v_pk_add_f32 %0, %1, %2\n") : "+v"(m) : "v"(m), "v"(m));
        if (OP == 13) asm volatile(REP64("v_cmp_eq_u32_e64 %0, %1, %2\n") : "+s"(m) : "v"(v), "v"(u));
        if (OP == 14) asm volatile(REP64("v_subrev_u32 %0, %1, %2\n") : "+v"(u) : "s"((unsigned)m), "v"(v));
        if (OP == 15) asm volatile(REP64("v_add_u32 %0, 0x5a0, %1\n") : "+v"(u) : "v"(v));
        if (OP == 16) asm volatile(REP64("v_xor_b32 %0, %1, %2\n") : "+v"(u) : "v"(v), "v"(b));
        if (OP == 17) asm volatile(REP64("v_bfe_i32 %0, %1, 3, 1\n") : "+v"(u) : "v"(v));
        if (OP == 18) asm volatile(REP64("v_addc_co_u32 %0, vcc, %1, %2, vcc\n") : "+v"(u) : "v"(v), "v"(v) : "vcc");
        if (OP == 19) asm volatile(REP64("v_max_f32 %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (OP == 20) asm volatile(REP64("s_bfe_u32 %0, %1, 0x12000b\n") : "+s"(v) : "s"(u) : "scc");
        if (OP == 21) asm volatile(REP64("s_and_b32 %0, %1, 0x7ff\n") : "+s"(v) : "s"(u) : "scc");
        if (OP == 22) asm volatile(REP64("s_nop 0\n"));
        if (OP == 23) asm volatile(REP64("s_cselect_b32 %0, %1, %0\n") : "+s"(v) : "s"(u) : "scc");
        if (OP == 24) asm volatile(REP8(REP8("v_add_f32 %0, %2, %3\n s_and_b32 %1, %1, 0x7ff\n")) : "+v"(a), "+s"(v) : "v"(b), "v"(c) : "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a == 123.456f || u == 0xdeadbeefu || m == 42ull) out[0] = 0;     // keep the results alive
}

template <int OP>
static void run(const char *name)
{
    unsigned long long *d;
    hipMalloc(&d, 8 * 1024);
    const int iters = 2000;
    printf("%-22s", name);
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd;          // one workgroup on one CU: waves dealt round-robin over the 4 SIMDs
        hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
        unsigned long long t;
        hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        printf("  W=%d: %5.2f cyc/inst/SIMD", waves_per_simd, (double)t / ((double)waves_per_simd * 64 * iters));
    }
    printf("\n");
    hipFree(d);
}

int main()
{
    run<0>("v_add_f32"); run<6>("v_min_f32"); run<19>("v_max_f32"); run<5>("v_med3_f32"); run<12>("v_pk_add_f32");
    run<1>("v_cndmask_e64 (sgpr)"); run<4>("v_cndmask_e32 (vcc)"); run<2>("v_cmp_eq_f32 -> sgpr"); run<3>("v_cmp_eq_f32 -> vcc");
    run<13>("v_cmp_eq_u32 -> sgpr"); run<7>("v_alignbit_b32"); run<8>("v_and_or_b32 (sgpr)"); run<9>("v_bitop3_b32 (sgpr)");
    run<10>("v_lshlrev_b32"); run<11>("v_min_u32"); run<14>("v_subrev_u32 (sgpr)"); run<15>("v_add_u32 literal"); run<16>("v_xor_b32");
    run<17>("v_bfe_i32"); run<18>("v_addc_co_u32");
    printf("scalar unit (per instruction and SIMD as above; a CU-wide unit shows as a cost that does not fall with W):\n");
    run<20>("s_bfe_u32"); run<21>("s_and_b32"); run<23>("s_cselect_b32"); run<22>("s_nop 0"); run<24>("v_add_f32 + s_and_b32 (pairs)");
    return 0;
}
