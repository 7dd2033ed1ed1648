// valu_rate.hip - issue rate of the sliced comparison's instruction kinds on one SIMD, by resident waves:
// v_bitop3 with three vector operands, with one scalar operand, v_xor with a scalar operand, s_bfe_i32, and the
// comparison's actual mix (mask made on the scalar unit, used by a vector instruction).  Prints cycles per wave-instruction
// per SIMD (s_memtime is a 100 MHz clock: the kernel is timed with hipEvents and the part's clock is taken from a plain
// loop of known length).  hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kIters = 4096, kUnroll = 32;

template <int kKind> __global__ __launch_bounds__(64) void rate_kernel(uint32_t *out, uint32_t seed)
{
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u, c = threadIdx.x * 5u, d = threadIdx.x * 7u;
    uint32_t s = __builtin_amdgcn_readfirstlane(seed);
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll / 4; ++u) {
            if (kKind == 0) {  // vector operands only, four independent chains
                asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %3 bitop3:0x96\n"
                             "v_bitop3_b32 %2, %2, %3, %0 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %1 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            } else if (kKind == 1) {  // one scalar operand
                asm volatile("v_bitop3_b32 %0, %0, %1, %4 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %4 bitop3:0x96\n"
                             "v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %4 bitop3:0x96" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s));
            } else if (kKind == 2) {  // v_xor (32-bit encoding) with a scalar operand
                asm volatile("v_xor_b32 %0, %4, %0\n v_xor_b32 %1, %4, %1\n v_xor_b32 %2, %4, %2\n v_xor_b32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s));
            } else if (kKind == 3) {  // scalar only
                uint32_t t0, t1, t2, t3;
                asm volatile("s_bfe_i32 %0, %4, 0x10003\n s_bfe_i32 %1, %4, 0x10005\n s_bfe_i32 %2, %4, 0x10007\n s_bfe_i32 %3, %4, 0x10009" : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3) : "s"(s) : "scc");
                asm volatile("" : : "s"(t0), "s"(t1), "s"(t2), "s"(t3));
            } else if (kKind == 4) {  // the comparison's mix: mask on the scalar unit, two vector users
                uint32_t t0, t1;
                asm volatile("s_bfe_i32 %4, %6, 0x10003\n v_xor_b32 %0, %4, %0\n s_bfe_i32 %5, %6, 0x10013\n v_bitop3_b32 %0, %0, %1, %5 bitop3:0xf6\n"
                             "s_bfe_i32 %4, %6, 0x10004\n v_xor_b32 %2, %4, %2\n s_bfe_i32 %5, %6, 0x10014\n v_bitop3_b32 %2, %2, %3, %5 bitop3:0xf6"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=s"(t0), "=s"(t1) : "s"(s) : "scc");
            } else {  // vector-only version of the same mix (masks already in vector registers)
                asm volatile("v_xor_b32 %0, %1, %0\n v_bitop3_b32 %0, %0, %1, %3 bitop3:0xf6\n v_xor_b32 %2, %3, %2\n v_bitop3_b32 %2, %2, %3, %1 bitop3:0xf6"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int kKind> static double run(int waves_per_simd, uint32_t *out, int n_cus)
{
    const int blocks = n_cus * 4 * waves_per_simd;  // 64-thread workgroups: one wave each, spread over the SIMDs by the dispatcher
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<kKind>, dim3(blocks), dim3(64), 0, 0, out, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<kKind>, dim3(blocks), dim3(64), 0, 0, out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    uint32_t *out;
    hipMalloc(&out, (size_t)n_cus * 4 * 16 * 64 * 4);
    const char *names[6] = {"v_bitop3 v,v,v", "v_bitop3 v,v,s", "v_xor s,v", "s_bfe_i32", "mix s_bfe+v_xor+s_bfe+v_bitop3", "mix masks in VGPRs (v_xor+v_bitop3)"};
    printf("%d CUs, %.2f GHz nominal; cycles per instruction and SIMD (scalar: per CU) at nominal clock\n", n_cus, ghz);
    for (int kind = 0; kind < 6; ++kind) {
        printf("%-40s", names[kind]);
        for (int wv : {1, 2, 4, 6, 8}) {
            double ms = 0;
            switch (kind) {
            case 0: ms = run<0>(wv, out, n_cus); break;
            case 1: ms = run<1>(wv, out, n_cus); break;
            case 2: ms = run<2>(wv, out, n_cus); break;
            case 3: ms = run<3>(wv, out, n_cus); break;
            case 4: ms = run<4>(wv, out, n_cus); break;
            default: ms = run<5>(wv, out, n_cus); break;
            }
            const double per_wave = (double)kIters * kUnroll * (kind == 4 ? 2 : 1);  // instructions per wave (mix: 8 per 4 "slots")
            const double cycles = ms * 1e-3 * ghz * 1e9;
            // per SIMD: wv waves; the scalar unit serves the 4 SIMDs of a CU
            const double per_instr = kind == 3 ? cycles / (per_wave * wv * 4) : cycles / (per_wave * wv);
            printf("  %dw %.2f", wv, per_instr);
        }
        printf("\n");
    }
    return 0;
}
