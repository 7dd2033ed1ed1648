// sort_probe.hip - experiment: rocPRIM onesweep throughput for the hit records as (u64 key, u32 value),
// (u64 key, u8 value) and u64 keys only; 40 sorted bits, n random records.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill(uint64_t *k, uint64_t n, int shift)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = (z >> (64 - 47)) << shift;
}

template <class V, class Config = rocprim::default_config> int run(const char *name, uint64_t n, unsigned begin, unsigned end, int shift)
{
    uint64_t *ka, *kb; V *va, *vb;
    CK(hipMalloc(&ka, n * 8)); CK(hipMalloc(&kb, n * 8)); CK(hipMalloc(&va, n * sizeof(V))); CK(hipMalloc(&vb, n * sizeof(V)));
    hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ka, n, shift);
    CK(hipMemset(va, 1, n * sizeof(V)));
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs<Config>((void *)nullptr, tb, ka, kb, va, vb, (size_t)n, begin, end));
    void *tmp; CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        CK(rocprim::radix_sort_pairs<Config>(tmp, tb, ka, kb, va, vb, (size_t)n, begin, end));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s n=%llu bits[%u,%u): %.2f ms\n", name, (unsigned long long)n, begin, end, ms);
    }
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(tmp);
    return 0;
}

int run_keys(uint64_t n, unsigned begin, unsigned end, int shift)
{
    uint64_t *ka, *kb;
    CK(hipMalloc(&ka, n * 8)); CK(hipMalloc(&kb, n * 8));
    hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ka, n, shift);
    size_t tb = 0;
    CK(rocprim::radix_sort_keys((void *)nullptr, tb, ka, kb, (size_t)n, begin, end));
    void *tmp; CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        CK(rocprim::radix_sort_keys(tmp, tb, ka, kb, (size_t)n, begin, end));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("keys only n=%llu bits[%u,%u): %.2f ms\n", (unsigned long long)n, begin, end, ms);
    }
    hipFree(ka); hipFree(kb); hipFree(tmp);
    return 0;
}

int main(int argc, char **argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1632068106ull;
    using namespace rocprim;
    if (run<uint32_t>("u64+u32 default", n, 7, 47, 0)) return 1;
#define CFG(HB, HI, SB, SI, BITS, ALG) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<SB, SI>, BITS, block_radix_rank_algorithm::ALG>>
    if (run<uint32_t, CFG(256, 12, 512, 12, 8, match)>("u64+u32 512x12 r8 match", n, 7, 47, 0)) return 1;
    if (run<uint32_t, CFG(256, 12, 256, 16, 8, match)>("u64+u32 256x16 r8 match", n, 7, 47, 0)) return 1;
    if (run<uint32_t, CFG(256, 12, 1024, 6, 8, match)>("u64+u32 1024x6 r8 match", n, 7, 47, 0)) return 1;
    if (run<uint32_t, CFG(256, 12, 512, 16, 8, match)>("u64+u32 512x16 r8 match", n, 7, 47, 0)) return 1;
    if (run<uint32_t, CFG(256, 12, 512, 12, 10, match)>("u64+u32 512x12 r10 match", n, 7, 47, 0)) return 1;
    if (run<uint32_t, CFG(256, 12, 1024, 8, 10, match)>("u64+u32 1024x8 r10 match", n, 7, 47, 0)) return 1;
    return 0;
}
