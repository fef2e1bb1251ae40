// Where does the dispatcher put the wavefronts of a workgroup?  For workgroups of W waves with enough
// dynamic LDS that exactly two fit on a CU, every wave records (XCC, SE, SH, CU, SIMD, wave slot) from
// HW_REG_HW_ID / HW_REG_XCC_ID.  The LDPC kernel's workgroup shape (6 active waves per frame) is decided
// on this: build with  hipcc --offload-arch=gfx950 -O2 tools/probe_placement.hip -o /tmp/probe  and run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(uint32_t *out, int spin)
{
    extern __shared__ float smem[];
    const int wave = threadIdx.x / 64;
    const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID, all 32 bits
    const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID [3:0]
    // keep every workgroup resident for a while so that the second one on a CU lands beside the first
    float a = (float)threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
    smem[threadIdx.x] = a;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + wave) * 2 + 0] = hw;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = xcc | (smem[(threadIdx.x + 1) % blockDim.x] > 1e30f ? 0x100u : 0u);
    }
}

// the same with 168 VGPRs per wave (3 waves fill a SIMD's register file): do two 6-wave workgroups still share a CU?
__global__ void __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(3, 3))) probe_fat(uint32_t *out, int spin)
{
    extern __shared__ float smem[];
    const int wave = threadIdx.x / 64;
    const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    float a = (float)threadIdx.x;
    asm volatile("v_mov_b32 v167, %0" : : "v"(a) : "v167");          // forces the 168-register allocation
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
    smem[threadIdx.x] = a;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + wave) * 2 + 0] = hw;
        out[(blockIdx.x * 16 + wave) * 2 + 1] = xcc | (smem[(threadIdx.x + 1) % blockDim.x] > 1e30f ? 0x100u : 0u);
    }
}

int main(int argc, char **argv)
{
    {
        const int lds = 79 * 1024, grid = 512, W = 6;
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe_fat), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        uint32_t *d;
        hipMalloc(&d, grid * 16 * 2 * 4);
        hipMemset(d, 0xFF, grid * 16 * 2 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe_fat, dim3(grid), dim3(W * 64), lds, 0, d, 2000000);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint32_t> h(grid * 16 * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        std::map<uint32_t, std::vector<int>> cu;
        for (int b = 0; b < grid; b++)
            for (int w = 0; w < W; w++) {
                const uint32_t hw = h[(b * 16 + w) * 2], xcc = h[(b * 16 + w) * 2 + 1] & 0xF;
                const uint32_t key = xcc << 16 | ((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15);
                if (!cu.count(key)) cu[key] = std::vector<int>(4, 0);
                cu[key][(hw >> 4) & 3]++;
            }
        std::map<std::vector<int>, int> hist;
        for (auto &kv : cu) hist[kv.second]++;
        printf("168-VGPR waves, 6-wave workgroups, 512 workgroups: %.1f ms (one round = all resident at once), waves per SIMD over the whole run:\n", ms);
        for (auto &kv : hist) printf("    %d+%d+%d+%d : %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
        hipLaunchKernelGGL(probe_fat, dim3(256), dim3(W * 64), lds, 0, d, 2000000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe_fat, dim3(256), dim3(W * 64), lds, 0, d, 2000000);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("  256 workgroups (one per CU): %.1f ms\n", ms);
        hipFree(d);
    }
    const int lds = 79 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int W : {6, 8, 12}) {
        const int grid = 512;
        uint32_t *d;
        hipMalloc(&d, grid * 16 * 2 * 4);
        hipMemset(d, 0xFF, grid * 16 * 2 * 4);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(W * 64), lds, 0, d, 200000);
        hipDeviceSynchronize();
        std::vector<uint32_t> h(grid * 16 * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        // per CU: waves per SIMD; per workgroup: waves per SIMD
        std::map<uint32_t, std::vector<int>> cu;      // key -> [simd0..3]
        std::map<std::vector<int>, int> wg_hist, cu_hist;
        for (int b = 0; b < grid; b++) {
            std::vector<int> per(4, 0);
            for (int w = 0; w < W; w++) {
                const uint32_t hw = h[(b * 16 + w) * 2], xcc = h[(b * 16 + w) * 2 + 1] & 0xF;
                const int simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const uint32_t key = xcc << 16 | se << 8 | sh << 4 | cuid;
                if (!cu.count(key)) cu[key] = std::vector<int>(4, 0);
                cu[key][simd]++;
                per[simd]++;
                if (b < 4) printf("W=%d wg %d wave %d: xcc %u se %d sh %d cu %d simd %d slot %u\n", W, b, w, xcc, se, sh, cuid, simd, hw & 15);
            }
            wg_hist[per]++;
        }
        for (auto &kv : cu) cu_hist[kv.second]++;
        printf("W=%d: %zu distinct CUs\n  waves-per-SIMD of a workgroup:\n", W, cu.size());
        for (auto &kv : wg_hist) printf("    %d+%d+%d+%d : %d workgroups\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
        printf("  waves-per-SIMD of a CU (all its workgroups):\n");
        for (auto &kv : cu_hist) printf("    %d+%d+%d+%d : %d CUs\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
        hipFree(d);
    }
    return 0;
}
