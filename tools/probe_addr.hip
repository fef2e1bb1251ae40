// Round 4: the circulant address of one slot, (4 t - shift) mod 1440 + base, as the layer loop forms it -- three 32-bit instructions + the base add -- against the same
// for TWO slots at once in packed 16-bit arithmetic (v_pk_sub_u16 / v_pk_add_u16 / v_pk_min_u16: the offset inside a row fits 16 bits) with the halves taken out by
// SDWA adds of the 32-bit base.  Checks every (t, shift) and times both forms on 1, 2, 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_addr.hip -o tools/bin/probe_addr && tools/bin/probe_addr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
__global__ void check(unsigned *bad, unsigned s0, unsigned s1, unsigned b0, unsigned b1)
{
    const unsigned t4 = threadIdx.x * 4u;
    if (t4 >= 1440u) return;
    const unsigned d0 = t4 - s0, d1 = t4 - s1;
    const unsigned r0 = min(d0, d0 + 1440u) + b0, r1 = min(d1, d1 + 1440u) + b1;
    const unsigned tt = t4 * 0x10001u, ss = s0 | (s1 << 16);
    unsigned d, e, w, a0, a1, k = 1440u * 0x10001u;
    asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"(tt), "s"(ss));
    asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(e) : "v"(d), "s"(k));
    asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(w) : "v"(d), "v"(e));
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(a0) : "s"(b0), "v"(w));
    asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(a1) : "s"(b1), "v"(w));
    if (a0 != r0 || a1 != r1) atomicAdd(bad, 1u);
}
template <int OP>
__global__ void __launch_bounds__(1024) probe(unsigned long long *out, unsigned *sink, int iters, unsigned s0, unsigned s1, unsigned b0, unsigned b1)
{
    unsigned t4 = (threadIdx.x & 63) * 4u, tt = t4 * 0x10001u, ss = s0 | (s1 << 16), k = 1440u * 0x10001u, acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        unsigned d, e, w, a0, a1;
        if (OP == 0)      // two slots, 32-bit: 8 instructions (+ 2 to keep the results alive)
            asm volatile(REP8("v_subrev_u32 %0, %5, %4\n v_add_u32 %1, 0x5a0, %0\n v_min_u32 %0, %0, %1\n v_add_u32 %2, %7, %0\n"
                              "v_subrev_u32 %0, %6, %4\n v_add_u32 %1, 0x5a0, %0\n v_min_u32 %0, %0, %1\n v_add_u32 %3, %8, %0\n v_xor_b32 %9, %9, %2\n v_xor_b32 %9, %9, %3\n")
                         : "=&v"(d), "=&v"(e), "=&v"(a0), "=&v"(a1) : "v"(t4), "s"(s0), "s"(s1), "s"(b0), "s"(b1), "v"(acc));
        else              // two slots, packed: 5 instructions (+ 2)
            asm volatile(REP8("v_pk_sub_u16 %0, %5, %6\n v_pk_add_u16 %1, %0, %7\n v_pk_min_u16 %2, %0, %1\n"
                              "v_add_u32_sdwa %3, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                              "v_add_u32_sdwa %4, %9, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_xor_b32 %10, %10, %3\n v_xor_b32 %10, %10, %4\n")
                         : "=&v"(d), "=&v"(e), "=&v"(w), "=&v"(a0), "=&v"(a1) : "v"(tt), "s"(ss), "s"(k), "s"(b0), "s"(b1), "v"(acc));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[threadIdx.x / 64] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main()
{
    unsigned *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    for (unsigned s0 = 0; s0 < 1440; s0 += 4) check<<<1, 384>>>(bad, s0, (s0 * 7 + 36) % 1440, 155520u, 1440u * (s0 % 100));
    unsigned hb = 1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("packed 16-bit circulant addresses: %u mismatches over 360 shifts x 360 lanes x 2 slots\n", hb);
    unsigned long long *d; unsigned *sink; hipMalloc(&d, 16 * 8); hipMalloc(&sink, 1024 * 4);
    const int iters = 4000;
    for (int op = 0; op < 2; op++)
        for (int waves : {1, 2, 3, 4}) {
            const int threads = 256 * waves;
            auto launch = [&] { if (op == 0) probe<0><<<1, threads>>>(d, sink, iters, 100u, 1000u, 155520u, 2880u); else probe<1><<<1, threads>>>(d, sink, iters, 100u, 1000u, 155520u, 2880u); };
            launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
            unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("%-40s %d wave(s) per SIMD: %6.2f ticks per PAIR of slots, %5.2f per instruction\n", op == 0 ? "32-bit (8 + 2 instructions per pair)" : "packed u16 + SDWA (5 + 2 per pair)", waves,
                   (double)h[0] / (8.0 * iters), (double)h[0] / (8.0 * iters) / (op == 0 ? 10 : 7));
        }
    return 0;
}
