// Latency of DEPENDENT vector instructions on gfx950, as one lone wave sees it: a serial recurrence (the L&R synchronizer's damped autocorrelation, the frame
// synchronizer's average over the frames) advances by one v_mul + one v_add per step, each waiting for the one before.  Cycles per dependent instruction for
// W = 1, 2, 4 waves on one SIMD (every wave its own chain): hipcc --offload-arch=gfx950 -O2 tools/probe_dep.hip -o tools/bin/probe_dep && tools/bin/probe_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, float *sink, int iters, float seed)
{
    float r = seed + threadIdx.x * 1e-3f, al = 0.999f, t = seed * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile(REP64("v_mul_f32 %0, %1, %0\n v_add_f32 %0, %0, %2\n") : "+v"(r) : "v"(al), "v"(t));      // the recurrence: 128 dependent instructions
        if (OP == 1) asm volatile(REP64("v_fma_f32 %0, %1, %0, %2\n v_fma_f32 %0, %1, %0, %2\n") : "+v"(r) : "v"(al), "v"(t)); // fused (another rounding)
        if (OP == 2) asm volatile(REP64("v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %1\n") : "+v"(r) : "v"(al), "v"(t));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main()
{
    unsigned long long *d; float *sink; hipMalloc(&d, 16 * 8); hipMalloc(&sink, 1024 * 4);
    const int iters = 2000;
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    const char *nm[3] = {"v_mul + v_add (dependent)", "v_fma x 2 (dependent)", "v_add x 2 (dependent)"};
    for (int op = 0; op < 3; op++)
        for (int waves : {1, 2, 4, 8}) {              // waves w, w + 4, ... share a SIMD: 256 lanes = one wave per SIMD
            const int threads = 64 * waves * 4 > 1024 ? 1024 : 64 * waves * 4;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] { if (op == 0) probe<0><<<1, threads>>>(d, sink, iters, 1.f); else if (op == 1) probe<1><<<1, threads>>>(d, sink, iters, 1.f); else probe<2><<<1, threads>>>(d, sink, iters, 1.f); };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("%-28s %d wave(s) per SIMD: %6.2f memtime ticks, %6.2f ns per dependent instruction (wall %.1f us for %d)\n", nm[op], threads / 256, (double)h[0] / (128.0 * iters),
                   ms * 1e6 / (128.0 * iters), ms * 1e3, 128 * iters);
        }
    printf("clock rate attribute: %d kHz\n", clk);
    return 0;
}
