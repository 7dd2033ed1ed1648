// valu_kinds.hip - issue cost (cycles per wave-instruction and SIMD, 8 resident waves per SIMD) of the vector instruction
// kinds the forest walk is made of (rf_predict_kernel, pair nodes): shifts, field extracts, fused shift-adds, literals.
// Same method as valu_rate.hip: four independent chains per wave, timed with events, nominal clock.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_kinds.hip -o /tmp/valu_kinds && /tmp/valu_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 4096, kUnroll = 32;

#define KINDS(X)                                                                                   \
    X(0, "v_xor_b32 v,v", "v_xor_b32 %0, %1, %0\n v_xor_b32 %1, %2, %1\n v_xor_b32 %2, %3, %2\n v_xor_b32 %3, %0, %3")                                   \
    X(1, "v_lshrrev_b32 imm", "v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %1, 3, %2\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %3, 3, %0")                    \
    X(2, "v_lshrrev_b32 v", "v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %1, %2, %1\n v_lshrrev_b32 %2, %3, %2\n v_lshrrev_b32 %3, %0, %3")                   \
    X(3, "v_and_b32 literal", "v_and_b32 %0, 0x3800, %1\n v_and_b32 %1, 0x3800, %2\n v_and_b32 %2, 0x3800, %3\n v_and_b32 %3, 0x3800, %0")               \
    X(4, "v_and_b32 inline", "v_and_b32 %0, 31, %1\n v_and_b32 %1, 31, %2\n v_and_b32 %2, 31, %3\n v_and_b32 %3, 31, %0")                               \
    X(5, "v_bfe_u32 v,v,1", "v_bfe_u32 %0, %1, %2, 1\n v_bfe_u32 %1, %2, %3, 1\n v_bfe_u32 %2, %3, %0, 1\n v_bfe_u32 %3, %0, %1, 1")                    \
    X(6, "v_bfe_u32 v,imm,imm", "v_bfe_u32 %0, %1, 9, 1\n v_bfe_u32 %1, %2, 9, 1\n v_bfe_u32 %2, %3, 9, 1\n v_bfe_u32 %3, %0, 9, 1")                    \
    X(7, "v_lshl_add_u32 v,3,v", "v_lshl_add_u32 %0, %1, 3, %0\n v_lshl_add_u32 %1, %2, 3, %1\n v_lshl_add_u32 %2, %3, 3, %2\n v_lshl_add_u32 %3, %0, 3, %3") \
    X(8, "v_alignbit_b32 v,v,v", "v_alignbit_b32 %0, %1, %2, %0\n v_alignbit_b32 %1, %2, %3, %1\n v_alignbit_b32 %2, %3, %0, %2\n v_alignbit_b32 %3, %0, %1, %3") \
    X(9, "v_and_or_b32 v,s,v", "v_and_or_b32 %0, %1, %4, %0\n v_and_or_b32 %1, %2, %4, %1\n v_and_or_b32 %2, %3, %4, %2\n v_and_or_b32 %3, %0, %4, %3") \
    X(10, "v_mul_u32_u24 10,v", "v_mul_u32_u24 %0, 10, %1\n v_mul_u32_u24 %1, 10, %2\n v_mul_u32_u24 %2, 10, %3\n v_mul_u32_u24 %3, 10, %0")           \
    X(11, "v_or3_b32", "v_or3_b32 %0, %1, %2, %0\n v_or3_b32 %1, %2, %3, %1\n v_or3_b32 %2, %3, %0, %2\n v_or3_b32 %3, %0, %1, %3")                     \
    X(12, "v_add3_u32", "v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %1, %2, %3, %1\n v_add3_u32 %2, %3, %0, %2\n v_add3_u32 %3, %0, %1, %3")                \
    X(13, "v_lshl_or_b32 v,1,v", "v_lshl_or_b32 %0, %1, 1, %0\n v_lshl_or_b32 %1, %2, 1, %1\n v_lshl_or_b32 %2, %3, 1, %2\n v_lshl_or_b32 %3, %0, 1, %3") \
    X(14, "v_add_u32 v,v", "v_add_u32 %0, %1, %0\n v_add_u32 %1, %2, %1\n v_add_u32 %2, %3, %2\n v_add_u32 %3, %0, %3")                                 \
    X(15, "v_mad_u32_u24 v,20,v", "v_mad_u32_u24 %0, %1, 20, %0\n v_mad_u32_u24 %1, %2, 20, %1\n v_mad_u32_u24 %2, %3, 20, %2\n v_mad_u32_u24 %3, %0, 20, %3") \
    X(16, "v_lshlrev_b32 imm", "v_lshlrev_b32 %0, 4, %1\n v_lshlrev_b32 %1, 4, %2\n v_lshlrev_b32 %2, 4, %3\n v_lshlrev_b32 %3, 4, %0")                  \
    X(17, "v_bfe_i32 v,v,1", "v_bfe_i32 %0, %1, %2, 1\n v_bfe_i32 %1, %2, %3, 1\n v_bfe_i32 %2, %3, %0, 1\n v_bfe_i32 %3, %0, %1, 1")                   \
    X(18, "v_and_b32 v,v", "v_and_b32 %0, %1, %0\n v_and_b32 %1, %2, %1\n v_and_b32 %2, %3, %2\n v_and_b32 %3, %0, %3")                                 \
    X(19, "v_bfi_b32 v,v,v", "v_bfi_b32 %0, %1, %2, %0\n v_bfi_b32 %1, %2, %3, %1\n v_bfi_b32 %2, %3, %0, %2\n v_bfi_b32 %3, %0, %1, %3")               \
    X(20, "v_perm_b32 v,v,v", "v_perm_b32 %0, %1, %2, %0\n v_perm_b32 %1, %2, %3, %1\n v_perm_b32 %2, %3, %0, %2\n v_perm_b32 %3, %0, %1, %3")          \
    X(21, "v_and_or_b32 v,v,v", "v_and_or_b32 %0, %1, %2, %0\n v_and_or_b32 %1, %2, %3, %1\n v_and_or_b32 %2, %3, %0, %2\n v_and_or_b32 %3, %0, %1, %3") \
    X(22, "v_bfe_u32 v,v,8", "v_bfe_u32 %0, %1, %2, 8\n v_bfe_u32 %1, %2, %3, 8\n v_bfe_u32 %2, %3, %0, 8\n v_bfe_u32 %3, %0, %1, 8")                   \
    X(23, "v_lshl_add_u32 v,11,v", "v_lshl_add_u32 %0, %1, 11, %0\n v_lshl_add_u32 %1, %2, 11, %1\n v_lshl_add_u32 %2, %3, 11, %2\n v_lshl_add_u32 %3, %0, 11, %3")

template <int kKind> __global__ __launch_bounds__(64) void rate_kernel(uint32_t *out)
{
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u, c = threadIdx.x * 5u, d = threadIdx.x * 7u;
    const uint32_t sc = __builtin_amdgcn_readfirstlane(out != nullptr ? 0x3800u : 0u);
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int u = 0; u < kUnroll / 4; ++u) {
#define X(n, name, text) if (kKind == n) asm volatile(text : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(sc));
            KINDS(X)
#undef X
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int kKind> static double run(int waves_per_simd, uint32_t *out, int n_cus)
{
    const int blocks = n_cus * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<kKind>, dim3(blocks), dim3(64), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<kKind>, dim3(blocks), dim3(64), 0, 0, out);
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
    printf("%d CUs, %.2f GHz nominal; cycles per wave-instruction and SIMD at 4 / 8 waves per SIMD\n", n_cus, ghz);
#define X(n, name, text)                                                                          \
    {                                                                                             \
        printf("%-26s", name);                                                                    \
        for (int wv : {4, 8}) {                                                                   \
            const double ms = run<n>(wv, out, n_cus);                                             \
            printf("  %dw %.2f", wv, ms * 1e-3 * ghz * 1e9 / ((double)kIters * kUnroll * wv));    \
        }                                                                                         \
        printf("\n");                                                                             \
    }
    KINDS(X)
#undef X
    return 0;
}
