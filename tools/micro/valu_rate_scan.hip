// valu_rate.hip - measures the issue rate of the integer VALU instructions the scan kernel is made of
// (v_xor_b32, v_bitop3_b32, v_bcnt_u32_b32, v_min3_u32) on gfx950, at 1/2/4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate_scan.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 13u;
    uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 11u;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_xor_b32 (VOP2)
            REP8(asm volatile("v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %3, %2\n v_xor_b32 %4, %5, %4\n v_xor_b32 %6, %7, %6\n"
                              "v_xor_b32 %1, %0, %1\n v_xor_b32 %3, %2, %3\n v_xor_b32 %5, %4, %5\n v_xor_b32 %7, %6, %7\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (KIND == 1) {  // v_bitop3_b32
            REP8(asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0xde\n v_bitop3_b32 %2, %3, %4, %2 bitop3:0xde\n"
                              "v_bitop3_b32 %4, %5, %6, %4 bitop3:0xde\n v_bitop3_b32 %6, %7, %0, %6 bitop3:0xde\n"
                              "v_bitop3_b32 %1, %0, %3, %1 bitop3:0xde\n v_bitop3_b32 %3, %2, %5, %3 bitop3:0xde\n"
                              "v_bitop3_b32 %5, %4, %7, %5 bitop3:0xde\n v_bitop3_b32 %7, %6, %1, %7 bitop3:0xde\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (KIND == 2) {  // v_bcnt_u32_b32
            REP8(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %4, %5, %4\n v_bcnt_u32_b32 %6, %7, %6\n"
                              "v_bcnt_u32_b32 %1, %0, %1\n v_bcnt_u32_b32 %3, %2, %3\n v_bcnt_u32_b32 %5, %4, %5\n v_bcnt_u32_b32 %7, %6, %7\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (KIND == 3) {  // v_min3_u32
            REP8(asm volatile("v_min3_u32 %0, %1, %2, %0\n v_min3_u32 %2, %3, %4, %2\n v_min3_u32 %4, %5, %6, %4\n v_min3_u32 %6, %7, %0, %6\n"
                              "v_min3_u32 %1, %0, %3, %1\n v_min3_u32 %3, %2, %5, %3\n v_min3_u32 %5, %4, %7, %5\n v_min3_u32 %7, %6, %1, %7\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (KIND == 4) {  // the scan mix: xor, bitop3, bcnt, (min3 every other)
            REP8(asm volatile("v_xor_b32 %0, %1, %2\n v_bitop3_b32 %0, %3, %0, %4 bitop3:0xde\n v_bcnt_u32_b32 %5, %0, 0\n"
                              "v_xor_b32 %0, %1, %3\n v_bitop3_b32 %0, %2, %0, %4 bitop3:0xde\n v_bcnt_u32_b32 %6, %0, 0\n"
                              "v_min3_u32 %7, %7, %5, %6\n v_xor_b32 %1, %1, %7\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
        } else if (KIND == 5) {  // v_xor with an SGPR operand
            uint32_t s = __builtin_amdgcn_readfirstlane(seed + i);
            REP8(asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %2, %8, %2\n v_xor_b32 %4, %8, %4\n v_xor_b32 %6, %8, %6\n"
                              "v_xor_b32 %1, %8, %1\n v_xor_b32 %3, %8, %3\n v_xor_b32 %5, %8, %5\n v_xor_b32 %7, %8, %7\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s));)
        } else if (KIND == 6) {  // v_bitop3 with an SGPR operand (as the compiler emits it)
            uint32_t s = __builtin_amdgcn_readfirstlane(seed + i);
            REP8(asm volatile("v_bitop3_b32 %0, %8, %2, %0 bitop3:0xde\n v_bitop3_b32 %2, %8, %4, %2 bitop3:0xde\n"
                              "v_bitop3_b32 %4, %8, %6, %4 bitop3:0xde\n v_bitop3_b32 %6, %8, %0, %6 bitop3:0xde\n"
                              "v_bitop3_b32 %1, %8, %3, %1 bitop3:0xde\n v_bitop3_b32 %3, %8, %5, %3 bitop3:0xde\n"
                              "v_bitop3_b32 %5, %8, %7, %5 bitop3:0xde\n v_bitop3_b32 %7, %8, %1, %7 bitop3:0xde\n"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

template <int KIND>
void run(const char *name, uint32_t *out, int cus)
{
    const int iters = 20000, per_iter = 64;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int groups = cus * wps;  // 256-thread groups = 4 waves = 1 wave per SIMD each
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(groups), dim3(256), 0, 0, out, 100, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(groups), dim3(256), 0, 0, out, iters, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double inst_per_simd = (double)iters * per_iter * wps;
        double cyc = ms * 1e-3 * 2.4e9;
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-instr/SIMD @2.4GHz  (%.1f T lane-ops/s)\n", name, wps, ms,
               cyc / inst_per_simd, inst_per_simd * 4 * cus * 64 / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, cus, p.clockRate);
    uint32_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    run<0>("v_xor_b32", out, cus);
    run<1>("v_bitop3_b32", out, cus);
    run<2>("v_bcnt_u32_b32", out, cus);
    run<3>("v_min3_u32", out, cus);
    run<4>("scan mix (xor,bitop3,bcnt,min3)", out, cus);
    run<5>("v_xor_b32 sgpr", out, cus);
    run<6>("v_bitop3_b32 sgpr", out, cus);
    return 0;
}
