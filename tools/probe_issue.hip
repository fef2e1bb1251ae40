// Round 4: what lets the waves of ONE SIMD issue side by side, and what does not.  W = 1 .. 4 waves per SIMD run the same unrolled stream:
//   0  VALU only            (v_subrev / v_add / v_min / v_add on registers)
//   1  + scalar operands    (s_and / s_bfe in front of every group, results read by the VALU instructions -- pass 1a of the LDPC layer without its loads)
//   2  + an LDS read        (ds_read_b32 per group, all waited for at the end -- pass 1a)
//   3  VALU + LDS read, no SALU
//   4  SALU only
// and report the s_memtime ticks EACH wave of SIMD 0 took (oldest first), so that starvation of the youngest wave shows.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_issue.hip -o tools/bin/probe_issue && tools/bin/probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, unsigned *sink, int iters, unsigned tabv)
{
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = (float)i;
    const unsigned t4 = (threadIdx.x & 63) * 4u;
    unsigned acc = 0, d, e, a; float v = 0.f; unsigned s3 = tabv, s4 = tabv >> 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile(REP16("v_subrev_u32 %0, %3, %4\n v_add_u32 %1, 0x5a0, %0\n v_min_u32 %0, %0, %1\n v_add_u32 %2, %3, %0\n") : "=&v"(d), "=&v"(e), "=&v"(a) : "v"(acc), "v"(t4));
        if (OP == 1) asm volatile(REP16("s_and_b32 %5, %6, 0x7ff\n v_subrev_u32 %0, %5, %4\n v_add_u32 %1, 0x5a0, %0\n s_bfe_u32 %5, %6, 0x12000b\n v_min_u32 %0, %0, %1\n v_add_u32 %2, %5, %0\n")
                                  : "=&v"(d), "=&v"(e), "=&v"(a) : "v"(acc), "v"(t4), "s"(s3), "s"(s4));
        if (OP == 2) asm volatile(REP16("s_and_b32 %5, %6, 0x7ff\n v_subrev_u32 %0, %5, %4\n v_add_u32 %1, 0x5a0, %0\n s_bfe_u32 %5, %6, 0x12000b\n v_min_u32 %0, %0, %1\n v_add_u32 %2, %5, %0\n ds_read_b32 %3, %4\n")
                                  "s_waitcnt lgkmcnt(0)\n" : "=&v"(d), "=&v"(e), "=&v"(a), "=&v"(v) : "v"(t4), "s"(s3), "s"(s4));
        if (OP == 3) asm volatile(REP16("v_subrev_u32 %0, %4, %4\n v_add_u32 %1, 0x5a0, %0\n v_min_u32 %0, %0, %1\n v_add_u32 %2, %4, %0\n ds_read_b32 %3, %4\n")
                                  "s_waitcnt lgkmcnt(0)\n" : "=&v"(d), "=&v"(e), "=&v"(a), "=&v"(v) : "v"(t4));
        if (OP == 4) asm volatile(REP16("s_and_b32 %0, %1, 0x7ff\n s_bfe_u32 %0, %1, 0x12000b\n s_and_b32 %0, %1, 0x7ff\n s_bfe_u32 %0, %1, 0x12000b\n") : "=&s"(s3) : "s"(s4));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
    if (threadIdx.x % 64 == 0) { out[threadIdx.x / 64] = t1 - t0; out[16 + threadIdx.x / 64] = (hw >> 4) & 3u; }
    sink[threadIdx.x] = acc + a + (unsigned)v + s3;
}
int main()
{
    unsigned long long *d; unsigned *sink; hipMalloc(&d, 32 * 8); hipMalloc(&sink, 1024 * 4);
    const int iters = 500;
    const char *nm[5] = {"VALU only (4 / group)", "VALU + SALU (4 + 2 / group)", "VALU + SALU + ds_read (4 + 2 + 1)", "VALU + ds_read (4 + 1)", "SALU only (4 / group)"};
    const int per[5] = {4, 6, 7, 5, 4};
    for (int op = 0; op < 5; op++)
        for (int waves : {1, 2, 3, 4}) {
            const int threads = 256 * waves;
            auto launch = [&] {
                if (op == 0) probe<0><<<1, threads>>>(d, sink, iters, 0x12345u); if (op == 1) probe<1><<<1, threads>>>(d, sink, iters, 0x12345u); if (op == 2) probe<2><<<1, threads>>>(d, sink, iters, 0x12345u);
                if (op == 3) probe<3><<<1, threads>>>(d, sink, iters, 0x12345u); if (op == 4) probe<4><<<1, threads>>>(d, sink, iters, 0x12345u); };
            launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
            unsigned long long h[32]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("%-36s %d wave(s) per SIMD; cycles per instruction of the waves on SIMD %llu:", nm[op], waves, h[16]);
            for (int w = 0; w < 4 * waves; w++) if (h[16 + w] == h[16]) printf(" %5.2f", (double)h[w] / (16.0 * iters * per[op]));
            printf("\n");
        }
    return 0;
}
