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

}  // namespace vsc
