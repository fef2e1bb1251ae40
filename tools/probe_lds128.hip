// Probe (GPU box): what a lone wave pays for ds_read_b128 / ds_write_b128 when every lane's 16 bytes sit in a row of its own (row stride S words) -- the chain wave of
// sync_metric_argmax_kernel<96> reads 24 such pieces per tick and waits ~1200 cycles for them.   hipcc --offload-arch=gfx950 -O3 tools/probe_lds128.hip -o tools/bin/probe_lds128
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int S, int NR, bool WR>
__global__ void __launch_bounds__(64) k(unsigned long long *out, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * S; i += 64) sm[i] = (float)i;
    __syncthreads();
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const f4 *src = reinterpret_cast<const f4 *>(&sm[lane * S]);
    f4 *dst = reinterpret_cast<f4 *>(&sm[lane * S]);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 200; it++) {
        f4 v[NR];
#pragma unroll
        for (int g = 0; g < NR; g++) v[g] = src[g];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NR; g++) acc += v[g];
        if (WR) {
#pragma unroll
            for (int g = 0; g < NR; g++) dst[g] = acc + (float)g;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = t1 - t0;
    sink[lane] = acc.x + acc.y + acc.z + acc.w;
}
template <int S, int NR, bool WR>
static void run(const char *what, unsigned long long *d, float *sink)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<S, NR, WR>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * S * 4);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<S, NR, WR>), dim3(1), dim3(64), 64 * S * 4, 0, d, sink);
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-46s row stride %3d words, %2d x b128 per trip%s: %7.1f ticks per trip, %6.1f per ds_read_b128\n", what, S, NR, WR ? " + as many writes" : "", h / 200.0, h / 200.0 / NR);
}
int main()
{
    unsigned long long *d; float *sink; hipMalloc(&d, 8); hipMalloc(&sink, 256);
    run<100, 24, false>("chain wave's pattern (100-word rows)", d, sink);
    run<100, 24, true>("chain wave's pattern with the writes", d, sink);
    run<28, 6, false>("24-frame rows (28 words)", d, sink);
    run<100, 6, false>("100-word rows, 6 reads", d, sink);
    run<100, 12, false>("100-word rows, 12 reads", d, sink);
    run<96, 24, false>("96-word rows (every lane on the same 4 banks)", d, sink);
    run<4, 1, false>("dense: lane i reads bytes 16 i .. 16 i + 15", d, sink);
    run<36, 8, false>("36-word rows", d, sink);
    run<132, 24, false>("132-word rows", d, sink);
    return 0;
}
