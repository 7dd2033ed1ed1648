// sort_probe2.hip - experiment: does rocPRIM sort (u32 key, u64 value) faster than (u64 key, u32 value)?
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class K> __global__ void fill(K *k, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = (K)(z >> 32);
}
template <class K, class V> int run(const char *name, uint64_t n, unsigned begin, unsigned end)
{
    K *ka, *kb; V *va, *vb;
    CK(hipMalloc(&ka, n * sizeof(K))); CK(hipMalloc(&kb, n * sizeof(K))); CK(hipMalloc(&va, n * sizeof(V))); CK(hipMalloc(&vb, n * sizeof(V)));
    hipLaunchKernelGGL(fill<K>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ka, n);
    CK(hipMemset(va, 1, n * sizeof(V)));
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs((void *)nullptr, tb, ka, kb, va, vb, (size_t)n, begin, end));
    void *tmp; CK(hipMalloc(&tmp, tb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        CK(rocprim::radix_sort_pairs(tmp, tb, ka, kb, va, vb, (size_t)n, begin, end));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 2) printf("%s n=%llu bits[%u,%u): %.2f ms\n", name, (unsigned long long)n, begin, end, ms);
    }
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(tmp);
    return 0;
}
int main(int argc, char **argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1632068106ull;
    if (run<uint32_t, uint64_t>("u32 key + u64 value", n, 0, 32)) return 1;
    if (run<uint32_t, uint32_t>("u32 key + u32 value", n, 0, 32)) return 1;
    if (run<uint64_t, uint32_t>("u64 key + u32 value", n, 0, 32)) return 1;
    return 0;
}
