// vsc_device.h - device helpers shared by the kernels of vsc_kernels.hip and vsc_seed.hip.
#pragma once

#include "vsc_internal.h"

namespace vsc {

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);  // ({hi,lo} >> sh)[31:0], sh in 0..31
}

// base + number of set bits of `mask` below this lane (v_mbcnt adds for free)
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, uint32_t base = 0u)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, base));
}

// The four waves of a workgroup never exchange data; lanes of one wave exchange data through their
// wave's LDS slice.  DS operations of one wave execute in order, so all that is needed is to stop
// the compiler from moving LDS accesses across the hand-off.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Workgroup barrier that also orders LDS accesses.  hipcc (ROCm 7.2, gfx950) leaves out the wait for outstanding
// LDS operations in front of s_barrier on some paths (seen in bin_hist_kernel, a loop back-edge into a barrier:
// one wave's ds_add landed after another wave had read and cleared the counter behind the barrier - 64 records
// missing from a histogram).  With the explicit wait every barrier is safe; use this instead of __syncthreads().
__device__ __forceinline__ void block_sync()
{
#ifndef VSC_PLAIN_SYNCTHREADS  // (defined only by tools/isa_barrier_excerpt.sh, which compiles - never runs - the plain form)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __syncthreads();
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// reverse complement of a 23-base plane: reverse the bit order, complement (A<->T, C<->G = NOT both planes)
__device__ __forceinline__ uint32_t revcomp_plane(uint32_t p) { return (~__brev(p)) >> 9; }
__device__ __forceinline__ uint32_t reverse23(uint32_t p) { return __brev(p) >> 9; }

// reads are fetched through the constant address space so that the (wave-uniform) loads are scalar
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4u *const_v4u_ptr;


// Is global position p the end (offset + length) of a contig?
__device__ __forceinline__ bool is_contig_end(const uint32_t *contig_end, uint32_t n_contigs, uint32_t p)
{
    uint32_t lo = 0, hi = n_contigs;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (contig_end[mid] < p) lo = mid + 1; else hi = mid;
    }
    return lo < n_contigs && contig_end[lo] == p;
}

// ------------------------------------------------------------------------------------------------
// packed feature rows: the 442 features of feature_matrix.h:25-126 in 64 bytes per hit (layout: include/varscot_hip.h,
// vsc_score_hits_packed); on / off = 23-base plane pairs of the on-target (the read) and the off-target in read orientation.
// Used by score_packed_kernel / rf_predict_kernel (vsc_kernels.hip) and by the rows-writing finalize (vsc_sort.hip).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void feature_row_packed(uint32_t on_h, uint32_t on_l, uint32_t off_h, uint32_t off_l, uint32_t (&w)[16])
{
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0;
    const uint32_t m21 = 0x1FFFFFu;
    const uint32_t mm = ((on_h ^ off_h) | (on_l ^ off_l)) & m21;  // positions 0..20 (:53)
    const uint32_t total = __popc(mm);
    const uint32_t adjacent = __popc(mm & (mm >> 1));             // :100-105
    const uint32_t seed = __popc(mm & 0xFFF00u);                  // 8 <= i <= 19, :94-98
    // transitions = mismatches whose codes differ in the hi bit only (AG, CT, GA, TC; :47)
    const uint32_t ts = __popc(mm & (on_h ^ off_h) & ~(on_l ^ off_l));
    uint32_t types = 0;
    for (uint32_t r = mm; r; r &= r - 1) {
        const int i = __ffs(r) - 1;
        const int o = (int)(((on_h >> i) & 1u) << 1 | ((on_l >> i) & 1u));
        const int b = (int)(((off_h >> i) & 1u) << 1 | ((off_l >> i) & 1u));
        types |= 1u << (o * 3 + (b > o ? b - 1 : b));             // :45-46,119
    }
    w[0] = mm | (total << 21) | (adjacent << 26);
    w[1] = types | (ts << 12) | ((total - ts) << 17) | (seed << 22);
    // (unrolled: bit 4 i + base lies in word 2 + i / 8, bit 16 i + pair in word 5 + i / 2 whatever the base is, so every word
    // index is a compile-time constant - no dynamically indexed register array)
#pragma unroll
    for (int i = 0; i < 21; ++i) {
        const uint32_t b = ((off_h >> i) & 1u) << 1 | ((off_l >> i) & 1u);
        w[2 + (i >> 3)] |= 1u << ((4 * (i & 7)) + b);             // :64-83
        if (i < 19) {                                             // :56-60
            const uint32_t b2 = ((off_h >> (i + 1)) & 1u) << 1 | ((off_l >> (i + 1)) & 1u);
            w[5 + (i >> 1)] |= 1u << (16 * (i & 1) + b * 4u + b2);
        }
    }
}

}  // namespace vsc
