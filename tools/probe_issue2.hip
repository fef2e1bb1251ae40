// Round 4, second issue probe: is the ~4.4 cycles a wave spends per vector instruction a DEPENDENCY latency or the SIMD's occupancy per wave64 instruction?
//   chain  : one register, every instruction depends on the one before
//   indep8 : eight registers, eight independent chains interleaved (a wave always has an instruction whose operands are ready)
// for v_add_f32, v_add_u32, v_min_u32 + literal (8-byte encoding) and v_cndmask with an SGPR mask; W = 1 .. 4 waves per SIMD; time of every wave of one SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_issue2.hip -o tools/bin/probe_issue2 && tools/bin/probe_issue2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, float *sink, int iters, float seed)
{
    float r0 = seed + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7, c = seed * 0.5f;
    unsigned long long m = 0x5555555555555555ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile(REP16("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n") : "+v"(r0) : "v"(c));
        if (OP == 1) asm volatile(REP16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        if (OP == 2) asm volatile(REP16("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c));
        if (OP == 3) asm volatile(REP16("v_add_u32 %0, 0x5a0, %0\n v_add_u32 %1, 0x5a0, %1\n v_add_u32 %2, 0x5a0, %2\n v_add_u32 %3, 0x5a0, %3\n v_add_u32 %4, 0x5a0, %4\n v_add_u32 %5, 0x5a0, %5\n v_add_u32 %6, 0x5a0, %6\n v_add_u32 %7, 0x5a0, %7\n")
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        if (OP == 4) asm volatile(REP16("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "s"(m));
        if (OP == 5) asm volatile(REP16("v_med3_f32 %0, %0, %8, %1\n v_min_f32 %1, %1, %8\n v_alignbit_b32 %2, %2, %3, 31\n v_and_or_b32 %3, %3, %9, %4\n v_med3_f32 %4, %4, %8, %5\n v_min_f32 %5, %5, %8\n v_alignbit_b32 %6, %6, %7, 31\n v_and_or_b32 %7, %7, %9, %0\n")
                                  : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "s"((unsigned)m));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    if (threadIdx.x % 64 == 0) { out[threadIdx.x / 64] = t1 - t0; out[16 + threadIdx.x / 64] = (hw >> 4) & 3u; }
    sink[threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}
int main()
{
    unsigned long long *d; float *sink; hipMalloc(&d, 32 * 8); hipMalloc(&sink, 1024 * 4);
    const int iters = 400;
    const char *nm[6] = {"v_add_f32 chain", "v_add_f32 indep8", "v_add_u32 indep8", "v_add_u32 literal indep8", "v_cndmask sgpr-mask indep8", "med3/min/alignbit/and_or mix"};
    for (int op = 0; op < 6; op++)
        for (int waves : {1, 2, 3, 4}) {
            const int threads = 256 * waves;
            auto launch = [&] {
                if (op == 0) probe<0><<<1, threads>>>(d, sink, iters, 1.f); if (op == 1) probe<1><<<1, threads>>>(d, sink, iters, 1.f); if (op == 2) probe<2><<<1, threads>>>(d, sink, iters, 1.f);
                if (op == 3) probe<3><<<1, threads>>>(d, sink, iters, 1.f); if (op == 4) probe<4><<<1, threads>>>(d, sink, iters, 1.f); if (op == 5) probe<5><<<1, threads>>>(d, sink, iters, 1.f); };
            launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
            unsigned long long h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("%-30s %d wave(s) per SIMD; cycles per instruction, waves of SIMD %llu:", nm[op], waves, h[16]);
            double mx = 0;
            for (int w = 0; w < 4 * waves; w++) if (h[16 + w] == h[16]) { printf(" %5.2f", (double)h[w] / (128.0 * iters)); if ((double)h[w] > mx) mx = (double)h[w]; }
            printf("   -> SIMD: one instruction per %.2f cycles\n", mx / (128.0 * iters) / waves);
        }
    return 0;
}
