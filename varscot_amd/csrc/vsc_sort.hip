// vsc_sort.hip - puts the hit records of a search in result order and assembles the vsc_hit records:
// the order of the reference's std::map per read and strand (VARSCOT_pipeline/read_mapping/
// bidir_mapping.cpp:154, key = (contig, position)) and rID / beginPos of every record (:99-100).
//
// The search kernels leave the hits of every region (64 reads) as packed 8-byte records in no particular
// order (layout: vsc_internal.h); ascending record order inside a region IS the result order.  A
// comparison-free "bin sort" in the shape of the memory system orders them, every record moving through
// HBM twice instead of once per 8-bit digit of a library radix sort:
//
//   bin_hist_kernel       one streaming read: records per (region, bin), bin = the top <= 11 key bits
//   bin_scan_kernel       bin starts (exclusive scan per region)
//   bin_partition_kernel  one read + one write: a workgroup takes a tile of 8 192 records, ranks them per bin
//                         with LDS atomics, reserves room in every bin with one coalesced returning atomic
//                         per bin, regroups the tile by bin in LDS and writes each bin's records as one
//                         contiguous piece.  Pieces of different tiles land in a bin in arbitrary order.
//   bin_finalize_kernel   one read + the 16-byte result write: resident workgroups take bins from a counter; a
//                         bin (<= 8 064 records, one read's hits on one strand inside a position window) is
//                         counting-sorted in LDS on the next <= 13 key bits, the handful of records that still
//                         agree are ranked (keys are unique), the contig is resolved and the vsc_hit records
//                         are written in place.
//
// A bin that exceeds the LDS capacity (a read with far more hits than the others - repeats) is listed and
// goes through another hist / scan / partition level on the following key bits; bins of a level are
// independent, so the levels are plain relaunches over a list of segments.  The streaming scan, whose hits
// arrive as (key, value) pairs with the full read index, enters through a level 0 that partitions by region
// and packs the records.
#include "vsc_internal.h"
#include "vsc_device.h"

namespace vsc {

namespace {

// Which record of its tile a thread takes as its k-th: two consecutive records per thread (one coalesced
// 16 bytes per lane) and pair of k.  Consecutive hits of a wave of the search kernel sit next to each other in a
// region and often share their bin (same read and strand, positions in the same slice); with one record per lane
// (record = k * threads + lane) the lanes of one LDS atomic carried runs of equal bins and the atomics serialised
// (hist + partition 9.0 -> 10.5 ms when strand-sorted buckets doubled the runs).  This way neighbours meet in
// different instructions: 9.3 ms.  (Four per lane 9.7, eight 12.5, sixteen 18.7: the loads stop coalescing.
// Scattering the records at the source instead - slot 37 i mod 128 of a block - brought the sort to 8.6 ms and
// the search kernel from 27 to 43 ms: its stores have to stay contiguous.)
constexpr int kLaneRun = 2;
__device__ __forceinline__ uint32_t tile_record(int k, uint32_t t)
{
    return (uint32_t)(k / kLaneRun) * (uint32_t)(kLaneRun * kSortThreads) + t * (uint32_t)kLaneRun + (uint32_t)(k % kLaneRun);
}

// Exclusive scan, in place, of the n (a power of two, <= 4 * kThreads) counters s[0..n); returns the total.
// Ends with a barrier; s_wave needs kThreads / 64 words.
template <int kThreads>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t *s, uint32_t n, uint32_t *s_wave)
{
    const uint32_t t = threadIdx.x, lane = t % kWave, wave = t / kWave;
    const uint32_t per = n > (uint32_t)kThreads ? n / kThreads : 1u;  // 1, 2 or 4
    const bool mine = t * per < n;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    uint32_t sum = 0;
    if (mine)
        for (uint32_t i = 0; i < per; ++i) {
            v[i] = s[t * per + i];
            sum += v[i];
        }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, kWave);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == kWave - 1) s_wave[wave] = inc;
    block_sync();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / kWave; ++w) {
        const uint32_t x = s_wave[w];
        before += (uint32_t)w < wave ? x : 0u;
        total += x;
    }
    uint32_t run = before + inc - sum;
    if (mine)
        for (uint32_t i = 0; i < per; ++i) {
            s[t * per + i] = run;
            run += v[i];
        }
    block_sync();
    return total;
}

// the segment that holds tile `tile`: last s with seg_tile0[s] <= tile (wave-uniform arguments: scalar loads)
__device__ __forceinline__ uint32_t segment_of_tile(const uint32_t *seg_tile0, uint32_t n_segs, uint32_t tile)
{
    uint32_t lo = 0, hi = n_segs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg_tile0[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

// level 0: a (key, value) pair of the streaming scan -> packed record (the read's region is the bin, not part of it)
__device__ __forceinline__ uint64_t pack_pair(uint64_t key, uint32_t val, uint32_t pos_pad, uint32_t pos_base)
{
    const uint64_t high = (key >> 32) & ((1ull << (kRecKeyBits - 32)) - 1ull);  // read inside its region, strand
    return (((high << 32) | (uint32_t)(((uint32_t)key - pos_base) << pos_pad)) << kRecPosShift) | (val & kMask23);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// histogram: records per (segment, bin)
// ------------------------------------------------------------------------------------------------
template <bool kPairs>
__global__ __launch_bounds__(kSortThreads) void bin_hist_kernel(const SortArgs a)
{
    __shared__ uint32_t s_hist[1 << kSortMaxBinBits];
    const uint32_t t = threadIdx.x;
    const uint32_t nbins = 1u << a.bin_bits;
    for (uint32_t b = t; b < nbins; b += kSortThreads) s_hist[b] = 0;
    block_sync();
    const uint32_t tile_begin = blockIdx.x * kHistTiles, tile_end = min(tile_begin + (uint32_t)kHistTiles, a.n_tiles);
    uint32_t seg = segment_of_tile(a.seg_tile0, a.n_segs, tile_begin);
    for (uint32_t tile = tile_begin; tile < tile_end; ++tile) {
        if (tile >= a.seg_tile0[seg + 1]) {
            // the tiles of this block cross into another segment: hand over what was counted so far
            block_sync();
            for (uint32_t b = t; b < nbins; b += kSortThreads) {
                const uint32_t c = s_hist[b];
                if (c) atomicAdd(&a.hist[((size_t)seg << a.bin_bits) + b], c);
                s_hist[b] = 0;
            }
            block_sync();
            seg = segment_of_tile(a.seg_tile0, a.n_segs, tile);
        }
        const SortSeg sg = a.segs[seg];
        const uint32_t first = (tile - a.seg_tile0[seg]) * (uint32_t)kSortTile;
        const uint32_t n = min(sg.n_in - first, (uint32_t)kSortTile);
        // all sixteen loads first, without branches (see bin_partition_kernel), then the counting
        uint64_t r[kSortItems];
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const uint32_t i = tile_record(k, t);
            const uint64_t at = sg.in_off + first + (i < n ? i : 0u);
            const uint64_t v = kPairs ? a.pair_keys[at] : a.in[at];
            r[k] = i < n ? v : kRecSentinel;
        }
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            if (kPairs) {
                if (tile_record(k, t) < n) atomicAdd(&s_hist[(uint32_t)(r[k] >> a.bin_shift)], 1u);
            } else {
                if (!(r[k] >> 63)) atomicAdd(&s_hist[(uint32_t)(r[k] >> a.bin_shift) & (nbins - 1u)], 1u);
            }
        }
    }
    block_sync();
    for (uint32_t b = t; b < nbins; b += kSortThreads) {
        const uint32_t c = s_hist[b];
        if (c) atomicAdd(&a.hist[((size_t)seg << a.bin_bits) + b], c);
    }
}

hipError_t launch_bin_hist(const SortArgs &args, hipStream_t stream)
{
    if (args.n_tiles == 0) return hipSuccess;
    const unsigned blocks = (args.n_tiles + kHistTiles - 1) / kHistTiles;
    if (args.pair_keys)
        hipLaunchKernelGGL(bin_hist_kernel<true>, dim3(blocks), dim3(kSortThreads), 0, stream, args);
    else
        hipLaunchKernelGGL(bin_hist_kernel<false>, dim3(blocks), dim3(kSortThreads), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// bin starts: one workgroup per segment
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void bin_scan_kernel(const SortArgs a)
{
    __shared__ uint32_t s_cnt[1 << kSortMaxBinBits];
    __shared__ uint32_t s_wave[512 / kWave];
    const uint32_t nbins = 1u << a.bin_bits;
    const size_t base = (size_t)blockIdx.x << a.bin_bits;
    for (uint32_t b = threadIdx.x; b < nbins; b += 512) s_cnt[b] = a.hist[base + b];
    block_sync();
    block_exclusive_scan<512>(s_cnt, nbins, s_wave);
    for (uint32_t b = threadIdx.x; b < nbins; b += 512) {
        a.bin_start[base + b] = s_cnt[b];
        if (!a.slot_cap) a.cursor[base + b] = s_cnt[b];  // (slot mode: hist IS the cursor array - the counts stay)
    }
}

hipError_t launch_bin_scan(const SortArgs &args, hipStream_t stream)
{
    if (args.n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(bin_scan_kernel, dim3(args.n_segs), dim3(512), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// partition: one tile of kSortTile records per workgroup
// ------------------------------------------------------------------------------------------------
// kSide: every record has a 32-bit side word in a parallel array (the site's lo plane of a search that keeps the sites'
// bases): it is moved with its record - through the same LDS buffer, in a second pass over the tile's slots.
template <bool kPairs, bool kSide = false>
__global__ __launch_bounds__(kSortThreads) void bin_partition_kernel(const SortArgs a)
{
    __shared__ uint64_t s_rec[kSortTile];
    __shared__ uint32_t s_cnt[(1 << kSortMaxBinBits) + kWave];  // records per bin, then their first slot in s_rec (+ spares)
    // Where this tile's records of a bin go in the bin: a thread keeps the answers of the bins it asked for in registers
    // until the tile is regrouped, then s_cnt[b] becomes "destination of slot i, less i" (a second table of 8 KB would
    // keep two workgroups from sharing a CU's LDS).  Level 0 (kPairs) looks bins up in the scanned counters: it keeps both.
    __shared__ uint32_t s_gbase[kPairs ? (1 << kSortMaxBinBits) : 1];
    __shared__ uint32_t s_over[(1 << kSortMaxBinBits) / 32 + 1];  // slot mode: bins that outgrew their slot (nothing is written for them); [last]: any
    __shared__ uint32_t s_wave[kSortThreads / kWave];
    const uint32_t t = threadIdx.x;
    const uint32_t nbins = 1u << a.bin_bits;
    // Workgroups are dealt round-robin over the 8 XCDs (workgroup b runs on XCD b % 8).  Each XCD takes a contiguous
    // eighth of the tiles, i.e. whole regions: the pieces different tiles append to one bin then meet in ONE L2,
    // which merges them into full lines before they leave for HBM.  (Speed only: nothing depends on the placement.)
    uint32_t tile = blockIdx.x;
    if (a.xcd_tiles) {
        tile = (blockIdx.x & 7u) * a.xcd_tiles + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= a.xcd_tiles || tile >= a.n_tiles) return;
    }
    const uint32_t seg = segment_of_tile(a.seg_tile0, a.n_segs, tile);
    const SortSeg sg = a.segs[seg];
    const uint32_t first = (tile - a.seg_tile0[seg]) * (uint32_t)kSortTile;
    const uint32_t n = min(sg.n_in - first, (uint32_t)kSortTile);
    for (uint32_t b = t; b < nbins + kWave; b += kSortThreads) s_cnt[b] = 0;
    for (uint32_t b = t; b < (1u << kSortMaxBinBits) / 32u + 1u; b += kSortThreads) s_over[b] = 0;
    block_sync();
    // records of the tile in registers; sentinels (and the slots past the tile) take no part.  Branch-free on
    // purpose (clamped loads + selects): with conditional stores into r[] / bin[] the compiler keeps the arrays
    // as 16-wide vectors, copies them at every branch and spills.
    uint64_t r[kSortItems];
    uint32_t bin[kSortItems];  // bin | rank inside (tile, bin) << 16
    uint32_t side[kSide ? kSortItems : 1];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = tile_record(k, t);
        const bool in_tile = i < n;
        const uint64_t at = sg.in_off + first + (in_tile ? i : 0u);
        if (kSide) side[k] = a.side_in[at];
        if (kPairs) {
            const uint64_t key = a.pair_keys[at];
            r[k] = in_tile ? pack_pair(key, a.pair_vals[at], a.pos_pad, a.pos_base) : kRecSentinel;
            bin[k] = in_tile ? (uint32_t)(key >> a.bin_shift) : 0u;
        } else {
            const uint64_t v = a.in[at];
            r[k] = in_tile ? v : kRecSentinel;
            bin[k] = (uint32_t)(r[k] >> a.bin_shift) & (nbins - 1u);
        }
    }
    // rank inside (tile, bin) = what the LDS atomic returns; sentinels add nothing to a spare counter - one per
    // lane: sentinels come in runs (the unused tail of a wave's block), and 64 lanes on ONE spare word would be
    // the slowest atomic of the kernel
    const uint32_t spare = nbins + (t % kWave);
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const bool real = !(r[k] >> 63);
        const uint32_t rank = atomicAdd(&s_cnt[real ? bin[k] : spare], real ? 1u : 0u);
        bin[k] |= rank << 16;  // a tile has <= 16 384 records: the rank fits 16 bits
    }
    block_sync();
    // room in the bins: consecutive lanes take consecutive bins (one coalesced returning atomic per 64 bins)
    constexpr int kOwnBins = (1 << kSortMaxBinBits) / kSortThreads;
    uint32_t own_at[kOwnBins];
#pragma unroll
    for (int j = 0; j < kOwnBins; ++j) {
        const uint32_t b = t + (uint32_t)j * kSortThreads;
        const uint32_t c = b < nbins ? s_cnt[b] : 0u;
        uint32_t at = c ? atomicAdd(&a.cursor[((size_t)seg << a.bin_bits) + b], c) : 0u;
        if (a.slot_cap && c && at + c > a.slot_cap) {  // the bin outgrew its slot: this level is repeated with a histogram
            atomicOr(a.overflow, 1u);
            atomicOr(&s_over[b >> 5], 1u << (b & 31u));
            s_over[(1u << kSortMaxBinBits) / 32u] = 1u;
        }
        own_at[j] = at;
        if (kPairs && b < nbins) s_gbase[b] = at;
    }
    const uint32_t total = block_exclusive_scan<kSortThreads>(s_cnt, nbins, s_wave);  // barriers inside
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t slot = s_cnt[bin[k] & 0xFFFFu] + (bin[k] >> 16);
        if (!(r[k] >> 63)) s_rec[slot] = r[k];
        if (kSide) bin[k] = slot;  // (the side word follows into the same slot, when s_cnt no longer holds the bins' first slots)
    }
    block_sync();
    if (!kPairs) {
        // slot i of the regrouped tile, a record of bin b, goes to record s_cnt[b] + i of the bin's place (the sum wraps)
#pragma unroll
        for (int j = 0; j < kOwnBins; ++j) {
            const uint32_t b = t + (uint32_t)j * kSortThreads;
            if (b < nbins) s_cnt[b] = own_at[j] - s_cnt[b];
        }
        block_sync();
    }
    const bool any_over = s_over[(1u << kSortMaxBinBits) / 32u] != 0u;
    // the tile, grouped by bin: neighbouring lanes write neighbouring records of a bin
    uint32_t slot_bin[kSide ? kSortItems / 2 : 1];  // kSide: the bin of every slot this thread writes, two to a word
    if (kSide) {
#pragma unroll
        for (int j = 0; j < kSortItems / 2; ++j) slot_bin[j] = 0;
    }
    for (uint32_t i = t; i < total; i += kSortThreads) {
        const uint64_t x = s_rec[i];
        // level 0 dropped the region from the record: find the bin of slot i in the scanned counters instead
        uint32_t b;
        if (kPairs) {
            uint32_t lo = 0, hi = nbins;  // last bin whose first slot is <= i (empty bins share a slot with their successor)
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_cnt[mid] <= i) lo = mid; else hi = mid;
            }
            b = lo;
        } else {
            b = (uint32_t)(x >> a.bin_shift) & (nbins - 1u);
        }
        const uint32_t in_bin = kPairs ? s_gbase[b] + (i - s_cnt[b]) : s_cnt[b] + i;
        const bool skip = any_over && ((s_over[b >> 5] >> (b & 31u)) & 1u);
        if (a.slot_cap) {
            if (!skip) a.out[(((uint64_t)seg << a.bin_bits) + b) * a.slot_cap + in_bin] = x;
        } else {
            a.out[sg.out_off + in_bin] = x;
        }
        if (kSide) {
            const uint32_t j = i / kSortThreads;  // this thread's j-th slot
#pragma unroll
            for (int q = 0; q < kSortItems / 2; ++q)
                if ((uint32_t)q == (j >> 1)) slot_bin[q] |= b << ((j & 1u) * 16u);
        }
    }
    if (kSide) {
        // the side words follow: into the LDS slots of their records (which have left), then out to the same places
        block_sync();
        uint32_t *const s_side = (uint32_t *)s_rec;
#pragma unroll
        for (int k = 0; k < kSortItems; ++k)
            if (!(r[k] >> 63)) s_side[bin[k]] = side[k];
        block_sync();
#pragma unroll
        for (int j = 0; j < kSortItems; ++j) {
            const uint32_t i = t + (uint32_t)j * kSortThreads;
            if (i < total) {
                const uint32_t b = (slot_bin[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                const uint32_t in_bin = s_cnt[b] + i;  // (kSide is never level 0)
                const bool skip = any_over && ((s_over[b >> 5] >> (b & 31u)) & 1u);
                if (a.slot_cap) {
                    if (!skip) a.side_out[(((uint64_t)seg << a.bin_bits) + b) * a.slot_cap + in_bin] = s_side[i];
                } else {
                    a.side_out[sg.out_off + in_bin] = s_side[i];
                }
            }
        }
    }
}

hipError_t launch_bin_partition(const SortArgs &args, hipStream_t stream)
{
    if (args.n_tiles == 0) return hipSuccess;
    const unsigned blocks = args.xcd_tiles ? 8u * args.xcd_tiles : args.n_tiles;
    if (args.pair_keys)
        hipLaunchKernelGGL(bin_partition_kernel<true>, dim3(blocks), dim3(kSortThreads), 0, stream, args);
    else if (args.side_in)
        hipLaunchKernelGGL((bin_partition_kernel<false, true>), dim3(blocks), dim3(kSortThreads), 0, stream, args);
    else
        hipLaunchKernelGGL(bin_partition_kernel<false>, dim3(blocks), dim3(kSortThreads), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// finalize: one bin per workgroup -> vsc_hit records in result order
// ------------------------------------------------------------------------------------------------
// Sub-bin counters / first slots are 16-bit (a bin has < 2^16 records), two to a word, so that 8 192
// sub-bins fit beside 8 000 records in HALF of the CU's LDS: the finer the sub-bins, the fewer records share
// one and have to be ranked against each other (0.6 per sub-bin on average at 5 000 records per bin), and with
// two workgroups per CU one computes while the other waits for its loads or drains its stores.
__device__ __forceinline__ uint32_t sub_slot(const uint32_t *s_sub, uint32_t sb)
{
    return (s_sub[sb >> 1] >> ((sb & 1u) * 16u)) & 0xFFFFu;
}

constexpr int kFinalizeAhead = 4;  // records of a sub-bin the ranking reads at once
constexpr int kFinalizeNear = 4;    // contigs a bin's position range may touch for the straight-line contig lookup
constexpr int kFinalizeRange = 64;  // ... for a binary search in LDS (more: in global memory - variant genomes)

// one bin (index `which` over all segments' bins), by the whole workgroup.
// kRows (a search that keeps the sites' bases, FinArgs.side_src / .rows): the record's low 23 bits and its side word are the
// site's hi / lo planes in read orientation; the mismatch mask is recomputed from them and the read, and the hit's 64-byte
// packed feature row (vsc_score_hits_packed's) is written beside its vsc_hit.  In LDS the record's plane bits give way to its
// load slot: the thread that ranks a record fetches the record and its side word again by that slot (from L2), so that no
// side word ever has to sit in LDS and hits and rows still leave in rank order, coalesced.
template <bool kRows>
__device__ __forceinline__ void finalize_bin(const FinArgs &a, const uint32_t which)
{
    __shared__ uint64_t s_rec[kSortCap];
    __shared__ uint32_t s_sub[(1 << (kRows ? kSortSubBitsRows : kSortSubBits)) / 2 + 1 + kWave];  // packed pairs of 16-bit counters, then first slots (+ spares)
    // kRows: 16 rows of 64 bytes per wave on their way out (see the end of the walk)
    __shared__ uint4 s_stage[kRows ? (kFinThreads / kWave) * 64 : 1];
    __shared__ uint32_t s_wave[kFinThreads / kWave];
    __shared__ uint32_t s_range[2 + kFinalizeRange];  // first contig of the bin's position range, contigs in it, their starts
    __shared__ uint32_t s_edge[4];                    // the bin's position range [0], [1]; contigs starting at or below either end [2], [3]
    // (opaque to the optimiser: hoisted out of the caller's loop over bins, the addresses derived from the thread
    // number stay live across it - 166 VGPRs wanted, 128 allowed for two workgroups per CU, the rest spilled)
    uint32_t t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    const uint32_t t = t_;
    const uint32_t seg = which >> a.bin_bits, bin = which & ((1u << a.bin_bits) - 1u);
    const SortSeg sg = a.segs[seg];
    uint64_t src = sg.in_off, dst = sg.final_off;
    uint64_t compact = sg.in_off;  // slot mode: where the bin would lie in the compact layout of the level's source buffer
    uint32_t n_src = sg.n_in;
    if (a.hist) {
        const size_t idx = ((size_t)seg << a.bin_bits) + bin;
        n_src = a.hist[idx];
        const uint32_t st = a.bin_start[idx];
        src = a.slot_cap ? (uint64_t)idx * a.slot_cap : sg.out_off + st;
        dst = sg.final_off + st;
        compact = sg.in_off + st;
    }
    if (n_src == 0) return;
    if (n_src > a.cap) {
        // too large to order in LDS: another partition level takes it (source and destination swap roles)
        if (t == 0) {
            const uint32_t slot = atomicAdd(a.n_over, 1u);
            if (slot < a.over_cap) {
                SortSeg o;
                o.in_off = src;
                o.out_off = a.slot_cap ? compact : src;  // (the next level writes into the buffer this level read)
                o.final_off = dst;
                o.n_in = n_src;
                o.guide_base = sg.guide_base;
                a.over[slot] = o;
            }
        }
        return;
    }
    const uint32_t nsub = 1u << a.sub_bits;
    const uint32_t nwords = nsub > 1u ? nsub / 2u : 1u;
    for (uint32_t i = t; i <= nwords + kWave; i += kFinThreads) s_sub[i] = 0;
    if (t < 4) s_edge[t] = 0;
    // a genome's contig table fits one entry per thread: loaded now, used after the counting pass (variant genomes
    // with millions of contigs take the binary searches further down)
    const bool few_contigs = a.n_contigs <= (uint32_t)kFinThreads;
    const uint32_t my_contig = (few_contigs && t < a.n_contigs) ? a.contig_off[t] : 0xFFFFFFFFu;
    block_sync();
    const uint64_t *const in = a.src + src;
    uint64_t r[kFinItems];
    uint32_t rk[kFinItems];
#pragma unroll
    for (int k = 0; k < kFinItems; ++k) {  // branch-free: see bin_partition_kernel
        const uint32_t i = k * kFinThreads + t;
        const uint64_t v = in[i < n_src ? i : 0u];
        r[k] = i < n_src ? v : kRecSentinel;
    }
    if (t == 0) {
        // the bin's position range: its keys agree in all bits above the sub-bin and rank fields (a sentinel - level 1
        // without a partition pass - has all bits set: the range then is everything, and so it is whenever those
        // fields cover the whole position)
        const uint32_t free_bits = a.sub_bits + a.low_bits;
        // (records hold positions relative to the shard's first one: the range is turned into global positions, saturating)
        const uint32_t any = (uint32_t)(r[0] >> kRecPosShift) >> a.pos_pad;
        const uint32_t p_lo = free_bits >= 32u ? 0u : (any >> free_bits) << free_bits;
        const uint64_t p_hi = free_bits >= 32u ? 0xFFFFFFFFull : (uint64_t)(p_lo | ((1u << free_bits) - 1u)) + a.pos_base;
        s_edge[0] = free_bits >= 32u ? 0u : (uint32_t)min((uint64_t)p_lo + a.pos_base, (uint64_t)0xFFFFFFFFull);
        s_edge[1] = (uint32_t)min(p_hi, (uint64_t)0xFFFFFFFFull);
    }
#pragma unroll
    for (int k = 0; k < kFinItems; ++k) {
        const bool real = !(r[k] >> 63);
        const uint32_t sb = (uint32_t)(r[k] >> a.sub_shift) & (nsub - 1u);
        const uint32_t sh = (sb & 1u) * 16u;
        // sentinels (and the slots past the bin) add nothing to a spare word of their own lane (one shared word
        // would serialise the 64 lanes of every instruction that lies wholly past the bin - a third of them).
        // (Skipping those instructions with a workgroup-uniform test made the compiler interleave loads and
        // atomics item by item: 13.9 -> 18.0 ms.)
        rk[k] = (atomicAdd(&s_sub[real ? sb >> 1 : nwords + 1u + (t % kWave)], real ? 1u << sh : 0u) >> sh) & 0xFFFFu;
    }
    block_sync();
    // contigs the range touches, all threads at once: one comparison per contig and thread, counted per wave
    // (thread 0 walking two binary searches through global memory here cost as much as loading the bin)
    if (few_contigs) {
        const uint32_t p_lo = s_edge[0], p_hi = s_edge[1];
        const bool mine = t < a.n_contigs;  // (the filler value of the other threads is a valid upper end of a range)
        const uint64_t le_lo = __ballot(mine && my_contig <= p_lo), le_hi = __ballot(mine && my_contig <= p_hi);
        if (t % kWave == 0) {
            if (le_lo) atomicAdd(&s_edge[2], (uint32_t)__popcll(le_lo));
            if (le_hi) atomicAdd(&s_edge[3], (uint32_t)__popcll(le_hi));
        }
    }
    // exclusive scan of the packed counters: a thread owns `per` consecutive words
    uint32_t n;
    {
        const uint32_t per = nwords > (uint32_t)kFinThreads ? nwords / kFinThreads : 1u;  // 1, 2, 4 or 8
        static_assert((1 << kSortSubBits) / 2 <= 8 * kFinThreads, "a thread scans at most 8 words of the sub-bin table");
        const bool mine = t * per < nwords;
        uint32_t w[8];
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            w[i] = (mine && (uint32_t)i < per) ? s_sub[t * per + i] : 0u;
            sum += (w[i] & 0xFFFFu) + (w[i] >> 16);
        }
        const uint32_t lane = t % kWave, wave = t / kWave;
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, kWave);
            if (lane >= (uint32_t)d) inc += o;
        }
        if (lane == kWave - 1) s_wave[wave] = inc;
        block_sync();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < kFinThreads / kWave; ++q) {
            const uint32_t x = s_wave[q];
            before += (uint32_t)q < wave ? x : 0u;
            total += x;
        }
        n = total;
        uint32_t run = before + inc - sum;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t lo = run;
            run += w[i] & 0xFFFFu;
            const uint32_t hi = run;
            run += w[i] >> 16;
            if (mine && (uint32_t)i < per) s_sub[t * per + i] = lo | (hi << 16);
        }
        // the end of the last sub-bin (slot nsub: the low half of the spare word, or the high half of the only word)
        if (t == 0 && nsub > 1u) s_sub[nwords] = n;
        if (few_contigs) {
            // every contig start at or below p_lo counts: the last of them is the range's first contig (start 0 exists)
            const uint32_t c_lo = max(s_edge[2], 1u) - 1u, c_hi = max(s_edge[3], 1u) - 1u;
            if (t == 0) {
                s_range[0] = c_lo;
                s_range[1] = c_hi - c_lo + 1u;
            }
            if (t >= c_lo && t < c_lo + (uint32_t)kFinalizeRange && t < a.n_contigs) s_range[2 + t - c_lo] = my_contig;
            // the slots past the last contig: a position no window has
            if (t < (uint32_t)kFinalizeRange && a.n_contigs - c_lo + t < (uint32_t)kFinalizeRange) s_range[2 + a.n_contigs - c_lo + t] = 0xFFFFFFFFu;
        } else if (t == 0) {
            const uint32_t p_lo = s_edge[0], p_hi = s_edge[1];
            uint32_t lo = 0, hi = a.n_contigs;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (a.contig_off[mid] <= p_lo) lo = mid; else hi = mid;
            }
            const uint32_t c_lo = lo;
            hi = a.n_contigs;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (a.contig_off[mid] <= p_hi) lo = mid; else hi = mid;
            }
            s_range[0] = c_lo;
            s_range[1] = lo - c_lo + 1u;
        }
        block_sync();
        // many contigs: the starts of the range's first contigs (absent: a position no window has)
        if (!few_contigs && t < (uint32_t)kFinalizeRange) s_range[2 + t] = t < s_range[1] ? a.contig_off[s_range[0] + t] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < kFinItems; ++k)
        if (!(r[k] >> 63))
            s_rec[sub_slot(s_sub, (uint32_t)(r[k] >> a.sub_shift) & (nsub - 1u)) + rk[k]] =
                kRows ? (r[k] & ~(uint64_t)kMask23) | (uint32_t)(k * kFinThreads + t) : r[k];  // kRows: the plane bits make room for the load slot
    block_sync();
    const uint32_t c_lo = s_range[0], c_n = s_range[1];
    const uint32_t st0 = s_range[2], st1 = s_range[3], st2 = s_range[4], st3 = s_range[5];
    if (!kRows) {
        // Two records per thread and round, every LDS read unconditional (clamped) so that independent reads leave
        // together: the phase is bound by LDS round trips (record -> sub-bin bounds -> the sub-bin's records), not by
        // bandwidth or issue.
        for (uint32_t base = 0; base < n; base += 2 * kFinThreads) {
            uint32_t s[2], e[2], smaller[2] = {0u, 0u};
            uint64_t x[2];
            bool live[2];
    #pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t idx = base + u * kFinThreads + t;
                live[u] = idx < n;
                const uint64_t v = s_rec[live[u] ? idx : 0u];
                x[u] = live[u] ? v : 0ull;
            }
    #pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t sb = (uint32_t)(x[u] >> a.sub_shift) & (nsub - 1u);
                const uint32_t first = sub_slot(s_sub, sb), next = sub_slot(s_sub, sb + 1u);
                s[u] = first;
                e[u] = live[u] ? next : first;
            }
            // rank among the records of the sub-bin (keys are unique): its first kFinalizeAhead records in one go, the
            // few longer sub-bins in a loop with a wave-uniform trip count (per-lane loops cost more in execution-mask
            // bookkeeping than they save)
            if (a.low_bits) {
                uint64_t y[2][kFinalizeAhead];
    #pragma unroll
                for (int u = 0; u < 2; ++u)
    #pragma unroll
                    for (int j = 0; j < kFinalizeAhead; ++j) y[u][j] = s_rec[min(s[u] + (uint32_t)j, (uint32_t)kSortCap - 1u)];
    #pragma unroll
                for (int u = 0; u < 2; ++u)
    #pragma unroll
                    for (int j = 0; j < kFinalizeAhead; ++j) smaller[u] += (uint32_t)(s[u] + (uint32_t)j < e[u] && y[u][j] < x[u]);
                for (uint32_t d = kFinalizeAhead;; ++d) {
                    const bool act0 = s[0] + d < e[0], act1 = s[1] + d < e[1];
                    if (__ballot(act0 || act1) == 0) break;
                    const uint64_t y0 = s_rec[act0 ? s[0] + d : 0u], y1 = s_rec[act1 ? s[1] + d : 0u];
                    smaller[0] += (uint32_t)(act0 && y0 < x[0]);
                    smaller[1] += (uint32_t)(act1 && y1 < x[1]);
                }
            }
    #pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!live[u]) continue;
                const uint32_t pos = ((uint32_t)(x[u] >> kRecPosShift) >> a.pos_pad) + a.pos_base;
                uint32_t c, start;
                if (c_n <= (uint32_t)kFinalizeNear) {
                    // the usual case: the bin's positions lie in at most four contigs - three comparisons
                    const uint32_t k = (uint32_t)(pos >= st1) + (uint32_t)(pos >= st2) + (uint32_t)(pos >= st3);
                    c = c_lo + k;
                    start = k == 0 ? st0 : (k == 1 ? st1 : (k == 2 ? st2 : st3));
                } else if (c_n <= (uint32_t)kFinalizeRange) {
                    uint32_t lo = 0, hi = c_n;  // last contig of the staged range whose start is <= pos
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_range[2 + mid] <= pos) lo = mid; else hi = mid;
                    }
                    c = c_lo + lo;
                    start = s_range[2 + lo];
                } else {
                    uint32_t lo = c_lo, hi = c_lo + c_n;
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (a.contig_off[mid] <= pos) lo = mid; else hi = mid;
                    }
                    c = lo;
                    start = a.contig_off[c];
                }
                const uint32_t mask = (uint32_t)x[u] & kMask23;
                uint4 h;
                h.x = sg.guide_base + ((uint32_t)(x[u] >> kRecReadShift) & (uint32_t)(kRegionReads - 1));  // vsc_hit.guide
                h.y = c;                                                                              // .contig
                h.z = pos - start;                                                                    // .pos
                h.w = ((uint32_t)(x[u] >> kRecStrandShift) & 1u) << 31 | (uint32_t)__popc(mask) << 23 | mask;  // .info
                // (written once, read by a later kernel at the earliest: past the caches - finalize 8.30 -> 8.15 ms)
                v4u hv;
                hv.x = h.x;
                hv.y = h.y;
                hv.z = h.z;
                hv.w = h.w;
                __builtin_nontemporal_store(hv, (v4u *)a.out + (dst + s[u] + smaller[u]));
            }
        }
    } else {
        // ---- the walk again, for records whose low 23 bits in LDS are their LOAD SLOT (the scatter above put it there): the
        // walker fetches the record and its side word again by that slot - from L2, the bin was loaded a moment ago - and has the
        // site's planes, from which the mismatch mask and the feature row follow; hit and row leave at bin start + rank,
        // neighbouring lanes writing neighbouring records and rows.
        const uint32_t *const side_in = a.side_src + src;
        const uint64_t kmask = ~(uint64_t)kMask23;
        for (uint32_t base = 0; base < n; base += 2 * kFinThreads) {
            uint32_t s[2], e[2], smaller[2] = {0u, 0u};
            uint64_t x[2], full[2];
            uint32_t lo_plane[2];
            bool live[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t idx = base + u * kFinThreads + t;
                live[u] = idx < n;
                const uint64_t v = s_rec[live[u] ? idx : 0u];
                x[u] = live[u] ? v : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {  // the gathers leave first, the ranking covers them
                const uint32_t slot = (uint32_t)x[u] & 0x1FFFu;
                full[u] = in[slot];
                lo_plane[u] = side_in[slot];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t sb = (uint32_t)(x[u] >> a.sub_shift) & (nsub - 1u);
                const uint32_t first = sub_slot(s_sub, sb), next = sub_slot(s_sub, sb + 1u);
                s[u] = first;
                e[u] = live[u] ? next : first;
            }
            if (a.low_bits) {
                uint64_t y[2][kFinalizeAhead];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < kFinalizeAhead; ++j) y[u][j] = s_rec[min(s[u] + (uint32_t)j, (uint32_t)kSortCap - 1u)];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < kFinalizeAhead; ++j) smaller[u] += (uint32_t)(s[u] + (uint32_t)j < e[u] && (y[u][j] & kmask) < (x[u] & kmask));
                for (uint32_t d = kFinalizeAhead;; ++d) {
                    const bool act0 = s[0] + d < e[0], act1 = s[1] + d < e[1];
                    if (__ballot(act0 || act1) == 0) break;
                    const uint64_t y0 = s_rec[act0 ? s[0] + d : 0u], y1 = s_rec[act1 ? s[1] + d : 0u];
                    smaller[0] += (uint32_t)(act0 && (y0 & kmask) < (x[0] & kmask));
                    smaller[1] += (uint32_t)(act1 && (y1 & kmask) < (x[1] & kmask));
                }
            }
            uint32_t w[2][16];
            uint64_t row_at[2] = {~0ull, ~0ull};  // where the row goes (rows of 64 bytes); ~0: no record
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!live[u]) continue;
                const uint32_t pos = ((uint32_t)(x[u] >> kRecPosShift) >> a.pos_pad) + a.pos_base;
                uint32_t c, start;
                if (c_n <= (uint32_t)kFinalizeNear) {
                    const uint32_t k = (uint32_t)(pos >= st1) + (uint32_t)(pos >= st2) + (uint32_t)(pos >= st3);
                    c = c_lo + k;
                    start = k == 0 ? st0 : (k == 1 ? st1 : (k == 2 ? st2 : st3));
                } else if (c_n <= (uint32_t)kFinalizeRange) {
                    uint32_t lo = 0, hi = c_n;  // last contig of the staged range whose start is <= pos
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_range[2 + mid] <= pos) lo = mid; else hi = mid;
                    }
                    c = c_lo + lo;
                    start = s_range[2 + lo];
                } else {
                    uint32_t lo = c_lo, hi = c_lo + c_n;
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (a.contig_off[mid] <= pos) lo = mid; else hi = mid;
                    }
                    c = lo;
                    start = a.contig_off[c];
                }
                const uint32_t strand = (uint32_t)(x[u] >> kRecStrandShift) & 1u;
                const uint32_t guide = sg.guide_base + ((uint32_t)(x[u] >> kRecReadShift) & (uint32_t)(kRegionReads - 1));
                const uint2 gp = a.guides[guide - a.guide_first];  // the read's planes: what the site was compared with
                const uint32_t hi_pl = (uint32_t)full[u] & kMask23, lo_pl = lo_plane[u] & kMask23;
                const uint32_t tm = ((hi_pl ^ gp.x) | (lo_pl ^ gp.y)) & kMask23;  // mismatches in read orientation ...
                const uint32_t mask = strand ? reverse23(tm) : tm;               // ... in forward-genome window coordinates
                v4u hv;
                hv.x = guide;
                hv.y = c;
                hv.z = pos - start;
                hv.w = strand << 31 | (uint32_t)__popc(mask) << 23 | mask;
                const uint64_t at = dst + s[u] + smaller[u];
                __builtin_nontemporal_store(hv, (v4u *)a.out + at);
                row_at[u] = at - a.rows_first;
                feature_row_packed(gp.x, gp.y, hi_pl, lo_pl, w[u]);
            }
            // The rows leave through the wave's kilobyte of LDS, sixteen at a time: sixteen lanes put their 64-byte rows
            // down, then every lane takes one 16-byte piece - four neighbouring lanes one whole row - so that a store
            // instruction writes sixteen complete 64-byte lines (streaming stores; a lane storing its own row, 16 bytes
            // per instruction 64 bytes apart, sends the lines out in quarters: 29 instead of 18 ms per c5 batch).
            uint4 *const stage = s_stage + (t / kWave) * 64;
            const uint32_t lane = t % kWave;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (__ballot(row_at[u] != ~0ull) == 0) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((lane >> 4) == (uint32_t)j) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)  // (piece q of row r in slot 4 r + (q ^ (r & 3)): the pieces of a row in different banks)
                            stage[4 * (lane & 15u) + ((uint32_t)q ^ (lane & 3u))] = make_uint4(w[u][4 * q], w[u][4 * q + 1], w[u][4 * q + 2], w[u][4 * q + 3]);
                    }
                    wave_sync();
                    const uint32_t r = lane >> 2, q = lane & 3u;  // this lane's piece: row r of the sixteen, quarter q
                    const uint32_t owner = 16u * (uint32_t)j + r;
                    const uint32_t at_lo = __shfl((uint32_t)row_at[u], (int)owner, kWave), at_hi = __shfl((uint32_t)(row_at[u] >> 32), (int)owner, kWave);
                    const uint4 piece = stage[4 * r + (q ^ (r & 3u))];
                    if ((at_lo & at_hi) != 0xFFFFFFFFu) {
                        v4u rv;
                        rv.x = piece.x;
                        rv.y = piece.y;
                        rv.z = piece.z;
                        rv.w = piece.w;
                        __builtin_nontemporal_store(rv, (v4u *)a.rows + ((((uint64_t)at_hi << 32) | at_lo) * 4 + q));
                    }
                    wave_sync();
                }
            }
        }
    }
}

// Two workgroups per CU stay resident and take bins from a counter.  One workgroup per bin, dealt round-robin over
// the XCDs in launch order, ran into the layout of the bins: a genome that fills 70 % of the 32-bit position space
// leaves the same five of every sixteen consecutive bins (one read's, one strand's) empty, always those of the same
// XCDs, and in-order dispatch makes the others wait for the busy ones (3 Gbp: 12.1 ms; 3.9 Gbp with 30 % more
// records: 11.2 ms; bins rotated by their group number: 11.1 ms).
// (amdgpu_waves_per_eu: two workgroups of 8 waves per CU = 4 waves per SIMD = at most 128 VGPRs; the loop makes the allocator ask for 166)
template <bool kRows>
__global__ __launch_bounds__(kFinThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void bin_finalize_kernel(const FinArgs a)
{
    __shared__ uint32_t s_next[2];
    if (a.overflow && *a.overflow) return;  // slot mode: the partition gave up - the level runs again, exactly
    const uint32_t total = a.n_segs << a.bin_bits;
    uint32_t which = blockIdx.x, parity = 0;
    while (which < total) {
        // the next bin's number is asked for now and looked at after this bin
        if (threadIdx.x == 0) s_next[parity] = atomicAdd(a.cursor, 1u) + gridDim.x;
        finalize_bin<kRows>(a, which);
        block_sync();  // the LDS tables are free again, s_next is written
        which = s_next[parity];
        parity ^= 1u;
    }
}

hipError_t launch_bin_finalize(const FinArgs &args, int max_groups, hipStream_t stream)
{
    if (args.n_segs == 0) return hipSuccess;
    const uint64_t bins = (uint64_t)args.n_segs << args.bin_bits;
    if (bins >= (1ull << 31) || args.sub_bits > (uint32_t)(args.rows ? kSortSubBitsRows : kSortSubBits) || args.bin_bits > (uint32_t)kSortMaxBinBits ||
        args.cap > (uint32_t)kSortCap || max_groups < 1)
        return hipErrorInvalidValue;
    const dim3 grid((unsigned)std::min<uint64_t>(bins, (uint64_t)max_groups));
    if (args.rows)
        hipLaunchKernelGGL(bin_finalize_kernel<true>, grid, dim3(kFinThreads), 0, stream, args);
    else
        hipLaunchKernelGGL(bin_finalize_kernel<false>, grid, dim3(kFinThreads), 0, stream, args);
    return hipGetLastError();
}

}  // namespace vsc
