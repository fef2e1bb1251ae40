// Round 4, third issue probe: SIMD throughput (one instruction per N cycles with 4 waves per SIMD, 8 independent registers per wave) and lone-wave cadence of the
// instruction classes the kernels of this library are made of -- the prices behind every "vector issue" fraction in profiles/ and DESIGN.md.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_issue3.hip -o tools/bin/probe_issue3 && tools/bin/probe_issue3
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define I8(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"
#define REGS "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, float *sink, int iters, float seed)
{
    float r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, c = seed * 0.5f;
    unsigned long long m = 0x5555555555555555ull; unsigned s1 = 3, s2 = 5;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile(REP16(I8("v_mul_f32", ", %8")) : REGS : "v"(c));
        if (OP == 1) asm volatile(REP16(I8("v_fma_f32", ", %8, %8")) : REGS : "v"(c));
        if (OP == 2) asm volatile(REP16(I8("v_exp_f32", "")) : REGS);
        if (OP == 3) asm volatile(REP16(I8("v_log_f32", "")) : REGS);
        if (OP == 4) asm volatile(REP16(I8("v_rcp_f32", "")) : REGS);
        if (OP == 5) asm volatile(REP16("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n")
                                  : "+v"(*(double *)&r0), "+v"(*(double *)&r2) : "v"(*(double *)&r4));
        if (OP == 6) asm volatile(REP16(I8("v_max_f32", ", |%8|")) : REGS : "v"(c));                  // VOP2 opcode pushed into the 64-bit encoding by a source modifier
        if (OP == 7) asm volatile(REP16(I8("v_add_f32", ", %8")) : REGS : "s"(s1));                   // VOP2 with an SGPR operand
        if (OP == 8) asm volatile(REP16("v_add_f32 %0, %0, %10\n s_add_u32 %8, %8, 1\n v_add_f32 %1, %1, %10\n s_add_u32 %9, %9, 1\n v_add_f32 %2, %2, %10\n s_add_u32 %8, %8, 1\n v_add_f32 %3, %3, %10\n s_add_u32 %9, %9, 1\n")
                                  : REGS, "+s"(s1), "+s"(s2) : "v"(c));                                 // VALU and SALU alternating: 4 + 4
        if (OP == 9) asm volatile(REP16(I8("v_cvt_f32_u32", "")) : REGS);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    if (threadIdx.x % 64 == 0) { out[threadIdx.x / 64] = t1 - t0; out[16 + threadIdx.x / 64] = (hw >> 4) & 3u; }
    sink[threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)(s1 + s2);
}
int main()
{
    unsigned long long *d; float *sink; hipMalloc(&d, 32 * 8); hipMalloc(&sink, 1024 * 4);
    const int iters = 300;
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char *nm[10] = {"v_mul_f32 (VOP2)", "v_fma_f32 (VOP3)", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_pk_add_f32 (VOP3P, 2 chains)", "v_max_f32 |abs| (VOP3 encoding)", "v_add_f32 with SGPR (VOP2)", "v_add_f32 + s_add_u32 alternating", "v_cvt_f32_u32 (VOP1)"};
    for (int op = 0; op < 8; op++)      // (8, VALU and SALU alternating, never finished on the GPU box -- the scalar operands of the asm block corrupt the loop counter --; 9 is left out with it)
        for (int waves : {1, 4}) {
            const int threads = 256 * waves;
            auto launch = [&] {
                switch (op) { case 0: probe<0><<<1, threads>>>(d, sink, iters, 1.f); break; case 1: probe<1><<<1, threads>>>(d, sink, iters, 1.f); break; case 2: probe<2><<<1, threads>>>(d, sink, iters, 1.f); break;
                              case 3: probe<3><<<1, threads>>>(d, sink, iters, 1.f); break; case 4: probe<4><<<1, threads>>>(d, sink, iters, 1.f); break; case 5: probe<5><<<1, threads>>>(d, sink, iters, 1.f); break;
                              case 6: probe<6><<<1, threads>>>(d, sink, iters, 1.f); break; case 7: probe<7><<<1, threads>>>(d, sink, iters, 1.f); break; case 8: probe<8><<<1, threads>>>(d, sink, iters, 1.f); break;
                              default: probe<9><<<1, threads>>>(d, sink, iters, 1.f); } };
            launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
            unsigned long long h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double mx = 0; int n = 0;
            for (int w = 0; w < 4 * waves; w++) if (h[16 + w] == h[16]) { if ((double)h[w] > mx) mx = (double)h[w]; n++; }
            const double per = 16.0 * 8 * iters;
            if (waves == 1) printf("%-36s lone wave: %5.2f cycles per instruction;", nm[op], (double)h[0] / per);
            else printf("  4 waves on a SIMD: one instruction per %5.2f cycles (slowest wave %5.2f per instruction)\n", mx / per / n, mx / per);
        }
    return 0;
}
