// sort_probe3.hip - experiment: 128 independent 3-pass sorts of 12.75 M (u64, u32) pairs each (one per read-range
// partition) against one 4-pass sort of all 1.63e9 pairs.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint64_t *k, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = z >> 17;
}
int main(int argc, char **argv)
{
    const uint64_t n = 1632068106ull;
    const int parts = argc > 1 ? atoi(argv[1]) : 128;
    uint64_t *ka, *kb; uint32_t *va, *vb;
    CK(hipMalloc(&ka, n * 8)); CK(hipMalloc(&kb, n * 8)); CK(hipMalloc(&va, n * 4)); CK(hipMalloc(&vb, n * 4));
    hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ka, n);
    CK(hipMemset(va, 1, n * 4));
    const uint64_t per = n / parts;
    size_t tb = 0, tb_all = 0;
    CK(rocprim::radix_sort_pairs((void *)nullptr, tb, ka, kb, va, vb, (size_t)per, 16u, 40u));
    CK(rocprim::radix_sort_pairs((void *)nullptr, tb_all, ka, kb, va, vb, (size_t)n, 15u, 47u));
    void *tmp; CK(hipMalloc(&tmp, tb > tb_all ? tb : tb_all));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        for (int p = 0; p < parts; ++p)
            CK(rocprim::radix_sort_pairs(tmp, tb, ka + p * per, kb + p * per, va + p * per, vb + p * per, (size_t)per, 16u, 40u));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%d sorts of %llu pairs, bits [16,40): %.2f ms\n", parts, (unsigned long long)per, ms);
    }
    CK(hipEventRecord(e0));
    CK(rocprim::radix_sort_pairs(tmp, tb_all, ka, kb, va, vb, (size_t)n, 15u, 47u));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one sort of all, bits [15,47): %.2f ms\n", ms);
    return 0;
}
