// Probe (GPU box): semantics of ds_read_addtid_b32 / ds_write_addtid_b32 on gfx950 -- LDS address = M0[15:0] + offset + TID * 4; is TID the lane (0..63) or the
// thread of the workgroup?   hipcc --offload-arch=gfx950 tools/probe_addtid.hip -o tools/bin/probe_addtid && tools/bin/probe_addtid
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = (float)i;
    __syncthreads();
    float v;
    const unsigned m0v = 16u * 4u;      // start at word 16
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_read_addtid_b32 %0 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "s"(m0v) : "m0", "memory");
    out[threadIdx.x] = v;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("thread 0 -> %g, thread 1 -> %g, thread 63 -> %g, thread 64 -> %g, thread 65 -> %g, thread 255 -> %g\n", h[0], h[1], h[63], h[64], h[65], h[255]);
    printf("(M0 = word 16, offset:8 = 2 words: lane-relative if thread 64 reads 18, workgroup-relative if it reads 82)\n");
    return 0;
}
