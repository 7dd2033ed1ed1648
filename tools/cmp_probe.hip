#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t v32u __attribute__((ext_vector_type(32)));

template <int T> __device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, T); }
__device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t &s, uint32_t &k) { s = bitop3<0x96>(a, b, c); k = bitop3<0xE8>(a, b, c); }
__device__ __forceinline__ void half_add(uint32_t a, uint32_t b, uint32_t &s, uint32_t &k) { s = a ^ b; k = a & b; }
__device__ __forceinline__ uint32_t spread(uint32_t w, int bit) { return (uint32_t)((int32_t)(w << (31 - bit)) >> 31); }
__device__ __forceinline__ void count7(const uint32_t *m, uint32_t &b0, uint32_t &b1, uint32_t &b2)
{
    uint32_t s1, c1, s2, c2, c3;
    full_add(m[0], m[1], m[2], s1, c1);
    full_add(m[3], m[4], m[5], s2, c2);
    full_add(s1, s2, m[6], b0, c3);
    full_add(c1, c2, c3, b1, b2);
}
__device__ __forceinline__ uint32_t tree(const uint32_t *mm, uint32_t budget, uint32_t valid)
{
    uint32_t a0, a1, a2, b0, b1, b2;
    count7(mm, a0, a1, a2);
    count7(mm + 7, b0, b1, b2);
    uint32_t t0, t1, t2, k0, k1, k2, j0, j1, j2, c0, c1, c2, c3, c4;
    full_add(a0, b0, mm[14], t0, k0);
    half_add(t0, mm[15], c0, j0);
    full_add(a1, b1, k0, t1, k1);
    half_add(t1, j0, c1, j1);
    full_add(a2, b2, k1, t2, k2);
    half_add(t2, j1, c2, j2);
    half_add(k2, j2, c3, c4);
    uint32_t le = ~c0 | spread(budget, 0);
    le = bitop3<0x8E>(c1, spread(budget, 1), le);
    le = bitop3<0x8E>(c2, spread(budget, 2), le);
    le = bitop3<0x8E>(c3, spread(budget, 3), le);
    uint32_t ok = bitop3<0x20>(le, c4, valid);
    // duplicate test on group A (as for segment 1)
    uint32_t d = ~a0 | spread(2u, 0);
    d = bitop3<0x8E>(a1, spread(2u, 1), d);
    d = bitop3<0x8E>(a2, spread(2u, 2), d);
    return ok & ~d;
}

// A: planes + spreads (the kernel's formulation)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_a(const uint32_t *sites, const uint32_t *reads, int n_reads, uint32_t *out)
{
    __shared__ uint32_t s_reads[1024];
    for (int i = threadIdx.x; i < n_reads; i += 256) s_reads[i] = reads[i];
    __syncthreads();
    uint32_t v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = sites[(blockIdx.x * 256 + threadIdx.x) * 32 + j];
    uint32_t acc = 0;
    for (int r = 0; r < n_reads; ++r) {
        const uint32_t rx = __builtin_amdgcn_readfirstlane(s_reads[r]);
        uint32_t mm[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) mm[q] = bitop3<0xF6>(v[q] ^ spread(rx, q), v[16 + q], spread(rx, 16 + q));
        acc ^= tree(mm, 6u + (rx & 1u), 0xFFFFFFFFu);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// B: one-hot planes, the read's base picks the register (uniform index -> relative VGPR addressing)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_b(const uint32_t *sites, const uint32_t *reads, int n_reads, uint32_t *out)
{
    __shared__ uint32_t s_reads[1024];
    for (int i = threadIdx.x; i < n_reads; i += 256) s_reads[i] = reads[i];
    __syncthreads();
    v32u e0, e1;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        e0[j] = sites[(blockIdx.x * 256 + threadIdx.x) * 64 + j];
        e1[j] = sites[(blockIdx.x * 256 + threadIdx.x) * 64 + 32 + j];
    }
    uint32_t acc = 0;
    for (int r = 0; r < n_reads; ++r) {
        const uint32_t rx = __builtin_amdgcn_readfirstlane(s_reads[r]);
        uint32_t mm[16];  // MISMATCH vectors: the one-hot planes are stored complemented
#pragma unroll
        for (int q = 0; q < 8; ++q) mm[q] = e0[4 * q + ((rx >> (2 * q)) & 3u)];
#pragma unroll
        for (int q = 0; q < 8; ++q) mm[8 + q] = e1[4 * q + ((rx >> (16 + 2 * q)) & 3u)];
        acc ^= tree(mm, 6u + (rx & 1u), 0xFFFFFFFFu);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// C: as A, but the budget test specialised for m = 8 (budget 8, 7 or 6 by a uniform branch) and the duplicate test for
// k = 2: constants folded, no SGPR operands in the tail of the tree
__device__ __forceinline__ uint32_t tree_m8(const uint32_t *mm, uint32_t d, uint32_t valid)
{
    uint32_t a0, a1, a2, b0, b1, b2;
    count7(mm, a0, a1, a2);
    count7(mm + 7, b0, b1, b2);
    uint32_t t0, t1, t2, k0, k1, k2, j0, j1, j2, c0, c1, c2, c3, c4;
    full_add(a0, b0, mm[14], t0, k0);
    half_add(t0, mm[15], c0, j0);
    full_add(a1, b1, k0, t1, k1);
    half_add(t1, j0, c1, j1);
    full_add(a2, b2, k1, t2, k2);
    half_add(t2, j1, c2, j2);
    half_add(k2, j2, c3, c4);
    uint32_t ok;
    if (d == 1) {  // count <= 7
        ok = bitop3<0x02>(c4, c3, valid);  // ~c4 & ~c3 & valid
    } else if (d == 0) {  // count <= 8: no bit 4, and not (bit 3 with anything below)
        const uint32_t low = c2 | c1 | c0;
        ok = bitop3<0x02>(c4, c3 & low, valid);
    } else {  // count <= 6: no bit 4, no bit 3, not 7
        const uint32_t seven = c2 & c1 & c0;
        ok = bitop3<0x02>(c4, c3 | seven, valid);
    }
    // duplicate test, k = 2: group A count <= 2  <=>  ~a2 & ~(a1 & a0)
    return ok & (a2 | (a1 & a0));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe_c(const uint32_t *sites, const uint32_t *reads, int n_reads, uint32_t *out)
{
    __shared__ uint32_t s_reads[1024];
    for (int i = threadIdx.x; i < n_reads; i += 256) s_reads[i] = reads[i];
    __syncthreads();
    uint32_t v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = sites[(blockIdx.x * 256 + threadIdx.x) * 32 + j];
    uint32_t acc = 0;
    for (int r = 0; r < n_reads; ++r) {
        const uint32_t rx = __builtin_amdgcn_readfirstlane(s_reads[r]);
        uint32_t mm[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) mm[q] = bitop3<0xF6>(v[q] ^ spread(rx, q), v[16 + q], spread(rx, 16 + q));
        acc ^= tree_m8(mm, rx % 3u, 0xFFFFFFFFu);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int blocks = 256 * 4 * 4, n_reads = 1000, reps = 20;
    std::vector<uint32_t> h((size_t)blocks * 256 * 64), hr(n_reads);
    uint32_t x = 12345;
    for (auto &w : h) { x = x * 1664525u + 1013904223u; w = x; }
    for (auto &w : hr) { x = x * 1664525u + 1013904223u; w = x; }
    uint32_t *d_sites, *d_reads, *d_out;
    hipMalloc(&d_sites, h.size() * 4); hipMalloc(&d_reads, hr.size() * 4); hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipMemcpy(d_sites, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_reads, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) {
                if (kind == 0) hipLaunchKernelGGL(probe_a, dim3(blocks), dim3(256), 0, 0, d_sites, d_reads, n_reads, d_out);
                else if (kind == 1) hipLaunchKernelGGL(probe_b, dim3(blocks), dim3(256), 0, 0, d_sites, d_reads, n_reads, d_out);
                else hipLaunchKernelGGL(probe_c, dim3(blocks), dim3(256), 0, 0, d_sites, d_reads, n_reads, d_out);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (w) {
                const double wave_reads = (double)blocks * 4 * n_reads * reps;
                std::printf("%s: %.3f ms per launch, %.1f cycles per wave-read per SIMD at 2.1 GHz (4 waves/SIMD)\n", kind == 2 ? "C specialised budget / duplicate tests" : kind ? "B one-hot + relative index" : "A planes + spreads",
                            ms / reps, ms * 1e-3 * 2.1e9 * 1024 / wave_reads);
            }
        }
    }
    return 0;
}
