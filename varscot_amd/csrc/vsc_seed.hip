// vsc_seed.hip - the seed-partitioned search: an exact pigeonhole pre-partition of the PAM-valid
// sites, kept resident in HBM, so that a read is only compared with the sites that can still be
// within its mismatch budget.
//
// The reference prunes with a pigeonhole split as well (two halves, floor(m/2) errors each,
// VARSCOT_pipeline/read_mapping/bidir_mapping.cpp:129-146,157-162) but walks an FM index for it.
// Here the 21 non-PAM read positions are cut into three 7-base segments.  A window within m
// mismatches of a read has at most k = floor(m/3) mismatches in at least one segment (otherwise it
// would have >= 3(k+1) > m).  Every PAM-valid site is filed three times, once per segment, under the
// 14-bit code of its 7 segment bases ("bucket").  A search enumerates, per read and segment, the
// 1 / 22 / 211 seven-mers within k substitutions of the read's segment, which turns into one list of
// reads per bucket; a wave then compares a chunk of one bucket's sites only with that bucket's reads:
// 3 * 211 / 16384 = 3.9 % of all (site, read) pairs at m = 8.  A pair is accepted on exactly the
// streaming scan's criterion - at most m mismatches over all 23 positions, the right-edge rule - so the
// accepted set is identical; a pair that qualifies in several segments is reported by the first only.
//
// Round 4: the cut is as tight as the PAM allows.  Inside a bucket the sites are grouped by their PAM (the "class":
// GG, GA or -P's), a chunk holds sites of one class, so the read's mismatches with the chunk's PAM are known before
// any site is looked at: they come out of the budget, read positions 21 and 22 leave the comparison, and the
// segments' thresholds follow what is LEFT for positions 0..20 (SeedPlan): a read that ends in GG meets the GA sites
// with one mismatch spent, so their third segment is searched within one substitution instead of two (22 buckets
// instead of 211) - 15 % fewer pairs at m = 8; at m = 6 thresholds like (1, 1, 2) / (1, 1, 1) or (0, 2, 2) / (0, 2, 1) stand
// in for (2, 2, 2): a quarter to two thirds of the pairs, or a whole table left unloaded - the host picks by cost.
//
// seed_sliced_kernel compares bit-sliced: 32 sites per lane and instruction, only the 14 positions of the other two
// segments are counted, the segment adds the list entry's known distance and the PAM the class's; hits are resolved
// from the site records and leave the kernel as packed 8-byte records (vsc_internal.h), one output region
// per 64 reads.
#include "vsc_internal.h"
#include "vsc_device.h"

#include <type_traits>

namespace vsc {

__device__ __forceinline__ uint32_t segment_key(uint32_t x, uint32_t l, int s)
{
    return (((x >> (kSegBases * s)) & 0x7Fu) << kSegBases) | ((l >> (kSegBases * s)) & 0x7Fu);
}

// ------------------------------------------------------------------------------------------------
// index build
// ------------------------------------------------------------------------------------------------
// (first PAM letter << 2 | second) of a 23-base plane pair in read orientation
__device__ __forceinline__ uint32_t pam_code(uint32_t x, uint32_t l)
{
    return (((x >> 21) & 1u) << 3) | (((l >> 21) & 1u) << 2) | (((x >> 22) & 1u) << 1) | ((l >> 22) & 1u);
}

// One 8-byte sort record per site and table: ((bucket << 2 | class) << 1 | strand) << 32 | index of the site.  The bin sort's
// partition kernels (vsc_sort.hip) order them by the 17 key bits in two levels (8 + 9 bits); the index then fetches the site.
__global__ __launch_bounds__(256) void seed_key_kernel(const uint4 *rec, uint64_t n, int seg, uint32_t pam_codes, uint32_t n_pam, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = rec[i];  // {hi plane | strand | edge, lo plane, position, 0}
    // class = the first PAM of the set the site ends in (the extraction only emits sites that end in one of them)
    const uint32_t code = pam_code(r.x, r.y);
    uint32_t cls = 0;
    for (uint32_t c = n_pam; c-- > 0;)
        if (((pam_codes >> (4 * c)) & 15u) == code) cls = c;
    // bucket, class, then strand: inside a (bucket, class) group the '+' sites precede the '-' sites, so that a chunk needs
    // one number (where its '-' sites begin) instead of a strand bit per site record
    const uint32_t key = (((segment_key(r.x, r.y, seg) << 2) | cls) << 1) | ((r.x >> kSiteStrandBit) & 1u);
    out[i] = ((uint64_t)key << 32) | (uint32_t)i;
}

// The extracted sites as 16-byte records {hi plane | strand | edge, lo plane, position, 0}: ordering a table by bucket
// then is ONE scattered 16-byte read per site (from three separate arrays it was three cache-line transactions per
// site: 55 ms per table, more than half of the index build).
__global__ __launch_bounds__(256) void seed_pack16_kernel(const uint32_t *x, const uint32_t *l, const uint32_t *pos, uint64_t n,
                                                          uint4 *rec)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i] = make_uint4(x[i], l[i], pos[i], 0u);
}

// the sites in bucket order: out[i] = rec[idx[i]] (what the bit-slicing pass and the compaction read)
__global__ __launch_bounds__(256) void seed_gather16_kernel(const uint4 *rec, const uint64_t *sorted, uint64_t n, uint4 *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = rec[(uint32_t)sorted[i]];  // the low word of a sort record is the site's index
}

hipError_t launch_seed_keys(const uint4 *rec, uint64_t n, int seg, uint32_t pam_codes, uint32_t n_pam, uint64_t *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rec, n, seg, pam_codes, n_pam, out);
    return hipGetLastError();
}

hipError_t launch_seed_pack16(const uint32_t *x, const uint32_t *l, const uint32_t *pos, uint64_t n, uint4 *rec, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_pack16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, l, pos, n, rec);
    return hipGetLastError();
}

hipError_t launch_seed_gather16(const uint4 *rec, const uint64_t *sorted, uint64_t n, uint4 *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_gather16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rec, sorted, n, out);
    return hipGetLastError();
}

// The resident site records of the hit path, 8 bytes: x = the 16 read positions OUTSIDE the table's segment
// (rest(hi) | rest(lo) << 16, the layout of a read-list entry), y = global position.  The segment's 7 bases
// are the bucket's code, the strand follows from the site's rank in its chunk (chunk table), and the rare
// "window is followed by N" flag lives in a bitmap that only chunks flagged in the chunk table look at.
__device__ __forceinline__ uint32_t rest_of(uint32_t v, uint32_t seg);

__global__ __launch_bounds__(256) void seed_compact_kernel(const uint4 *sites16, uint64_t n_per_table, uint64_t n, uint2 *sites8,
                                                           uint32_t *edge_bits)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = sites16[i];
    const uint32_t seg = (uint32_t)(i / n_per_table);
    sites8[i] = make_uint2(rest_of(r.x, seg) | (rest_of(r.y, seg) << 16), r.z);
    if ((r.x >> kSiteEdgeBit) & 1u) atomicOr(&edge_bits[i >> 5], 1u << (i & 31u));
}

// chunk_tab[c].z |= 1 << 28 where the chunk holds a site whose window is followed by N
__global__ __launch_bounds__(256) void seed_chunk_flags_kernel(uint4 *chunk_tab, uint32_t n_chunks, const uint32_t *edge_bits)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 ct = chunk_tab[c];
    if (ct.y == 0) return;
    const uint32_t first = ct.x, last = ct.x + ct.y - 1;
    uint32_t any = 0;
    for (uint32_t w = first >> 5; w <= last >> 5; ++w) {
        uint32_t bits = edge_bits[w];
        if (w == first >> 5) bits &= 0xFFFFFFFFu << (first & 31u);
        if (w == last >> 5) bits &= 0xFFFFFFFFu >> (31u - (last & 31u));
        any |= bits;
    }
    if (any) chunk_tab[c].z = ct.z | (1u << kChunkEdgeBit);
}

hipError_t launch_seed_compact(const uint4 *sites16, uint64_t n_per_table, uint64_t n, uint2 *sites8, uint32_t *edge_bits,
                               hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_compact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sites16, n_per_table, n, sites8,
                       edge_bits);
    return hipGetLastError();
}

hipError_t launch_seed_chunk_flags(uint4 *chunk_tab, uint32_t n_chunks, const uint32_t *edge_bits, hipStream_t stream)
{
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_chunk_flags_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, chunk_tab, n_chunks, edge_bits);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// per-search: neighbourhood enumeration and the per-bucket read lists
// ------------------------------------------------------------------------------------------------
// the 21 position pairs (p < q) of a 7-base segment
__constant__ uint8_t kPairP[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
__constant__ uint8_t kPairQ[21] = {1, 2, 3, 4, 5, 6, 2, 3, 4, 5, 6, 3, 4, 5, 6, 4, 5, 6, 5, 6, 6};

__device__ __forceinline__ void substitute(uint32_t &h, uint32_t &l, uint32_t p, uint32_t alt)
{
    const uint32_t c = (((h >> p) & 1u) << 1) | ((l >> p) & 1u);
    const uint32_t d = (c + alt) & 3u;  // alt in 1..3: the three other bases
    h = (h & ~(1u << p)) | ((d >> 1) << p);
    l = (l & ~(1u << p)) | ((d & 1u) << p);
}

// The per-bucket read lists are a counting sort over the 49 152 buckets, done by two passes of the same
// enumeration: pass 1 counts the entries of every bucket, a one-workgroup scan turns the counts into padded list
// starts, pass 2 recomputes every entry and drops it into its bucket's next free slot.  (Rounds 1-2 materialised
// (bucket, read) pairs and sorted them with rocPRIM: three launches x seven digit passes inside the timed step.)
// The order of the entries INSIDE a bucket's list is the order in which pass 2's atomics land - it decides nothing:
// every read of a list is compared with every site of the bucket, and the result is sorted afterwards.
//
// Neighbour numbering: 0 = the segment itself; 1..21 = one substitution (position * 3 + alt);
// 22..210 = two substitutions (pair * 9 + alt1 * 3 + alt2).  n_nbr = 1, 22 or 211 for k = 0, 1, 2.
__device__ __forceinline__ uint32_t seed_neighbour(const uint2 gp, uint32_t s, uint32_t n, uint32_t *dist)
{
    uint32_t h = (gp.x >> (kSegBases * s)) & 0x7Fu, l = (gp.y >> (kSegBases * s)) & 0x7Fu;
    if (n >= 22) {
        const uint32_t e = n - 22, pi = e / 9, alts = e % 9;
        substitute(h, l, kPairP[pi], alts / 3 + 1);
        substitute(h, l, kPairQ[pi], alts % 3 + 1);
    } else if (n >= 1) {
        const uint32_t e = n - 1;
        substitute(h, l, e / 3, e % 3 + 1);
    }
    // substitutions always change the base, so the read's own segment differs from this neighbour in
    // exactly d = 0 / 1 / 2 positions: the seed distance of every site filed under the neighbour's bucket
    *dist = n >= 22 ? 2u : (n >= 1 ? 1u : 0u);
    return s * kBucketsPerSeg + ((h << kSegBases) | l);
}

// The 16 read positions outside segment `seg`, packed in ascending order ("rest" of a 23-bit plane).
__device__ __forceinline__ uint32_t rest_of(uint32_t v, uint32_t seg)
{
    v &= kMask23;
    if (seg == 0) return v >> kSegBases;
    if (seg == 1) return (v & 0x7Fu) | ((v >> (2 * kSegBases)) << kSegBases);
    return (v & 0x3FFFu) | ((v >> (3 * kSegBases)) << (2 * kSegBases));
}

// One thread per (read, segment, neighbour).  kScatter = false: count[list]++.  kScatter = true: the list entry goes to
// poff[list] + (cursor[list]++) - x = rest(hi) | rest(lo) << 16, y = read index | per class: budget << (16 + 4 c).
// budget = what the comparison may still spend on the 14 positions of the other two segments against a site of class c:
// max_mm - mismatches of the read's last two letters with the class's PAM - seed distance (15: the read has nothing to do
// with that class).  Segments 0 and 1 keep ONE list per bucket, whose entries serve every class; segment 2 a list per
// (class, bucket), because which of its neighbours a read visits depends on what the class leaves it (SeedPlan).
template <bool kScatter>
__global__ __launch_bounds__(256) void seed_enum_kernel(const uint2 *guides, uint32_t n_guides, const SeedPlan plan, uint32_t *count,
                                                        const uint32_t *poff, uint2 *list_rest)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)n_guides * kSegments * plan.n_nbr;
    if (i >= total) return;
    const uint32_t n = (uint32_t)(i % plan.n_nbr);
    const uint32_t s = (uint32_t)((i / plan.n_nbr) % kSegments);
    const uint32_t g = (uint32_t)(i / ((uint64_t)plan.n_nbr * kSegments));
    const uint2 gp = guides[g];
    uint32_t d;
    const uint32_t b = seed_neighbour(gp, s, n, &d);
    const uint32_t mine = pam_code(gp.x, gp.y);
    const uint32_t rest = rest_of(gp.x, s) | (rest_of(gp.y, s) << 16);
    constexpr uint32_t kNone = (kListNoBudget << kListBudgetShift) | (kListNoBudget << (kListBudgetShift + 4)) |
                               (kListNoBudget << (kListBudgetShift + 8)) | (kListNoBudget << (kListBudgetShift + 12));
    uint32_t shared = kNone;  // segments 0 and 1: the budgets of all classes in one entry
    for (uint32_t c = 0; c < plan.n_pam; ++c) {
        const uint32_t diff = mine ^ ((plan.pam_codes >> (4 * c)) & 15u);
        const uint32_t spent = ((diff & 12u) ? 1u : 0u) + ((diff & 3u) ? 1u : 0u);
        if (spent + d > plan.max_mm) continue;
        const uint32_t left = plan.max_mm - spent;  // for read positions 0..20
        const int thr = s == 0 ? (int)plan.k0 : (s == 1 ? (int)plan.k1 : (plan.tight ? (int)left - (int)plan.k0 - (int)plan.k1 - 2 : (int)plan.k0));
        if ((int)d > thr) continue;
        const uint32_t field = kListBudgetShift + 4 * c;
        if (s < 2) {
            shared = (shared & ~(15u << field)) | ((left - d) << field);
            continue;
        }
        const uint32_t list = b + c * (uint32_t)kBucketsPerSeg;  // b = 2 x 16384 + code
        const uint32_t at = atomicAdd(&count[list], 1u);
        if (kScatter) list_rest[poff[list] + at] = make_uint2(rest, g | ((kNone & ~(15u << field)) | ((left - d) << field)));
    }
    if (s < 2 && shared != kNone) {
        const uint32_t at = atomicAdd(&count[b], 1u);
        if (kScatter) list_rest[poff[b] + at] = make_uint2(rest, g | shared);
    }
}

// poff[b] = sum over b' < b of roundup4(count[b']); the counts are cleared for pass 2; one workgroup, kLists + 1 outputs.
// A wave owns 5 120 consecutive lists: 80 coalesced rows of 64, read twice (the second time from L2): first for the wave's
// total - the waves' totals meet in LDS once -, then row by row, eight in flight: every row is scanned across the lanes by
// six DPP adds (no LDS), the rows' totals are chained on the scalar unit.  (Round 3's form - a thread summing its 48 consecutive counters one dependent, uncoalesced load after the
// other - took 0.17 ms: most of what a 1 000-read search spends on its lists.)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    // row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes (lanes without a source add 0), then lane 15 of rows 0 and 2 into
    // rows 1 and 3 (row_bcast:15, rows 0b1010), then lane 31 into rows 2 and 3 (row_bcast:31, rows 0b1100)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

__global__ __launch_bounds__(1024) void seed_pad_scan_kernel(uint32_t *count, uint32_t *poff)
{
    constexpr int kWaves = 1024 / kWave, kPerWave = kLists / kWaves, kRows = kPerWave / kWave;  // 16, 5 120, 80
    static_assert(kLists % (kWaves * kWave) == 0 && kRows % 8 == 0, "whole rows per wave, eight at a time");
    __shared__ uint32_t s_total[kWaves];
    const uint32_t lane = threadIdx.x % kWave, wave = uniform(threadIdx.x / kWave);
    uint32_t *const c = count + wave * kPerWave + lane;
    auto padded = [](uint32_t n) { return (n + (uint32_t)(kGuideUnroll - 1)) & ~(uint32_t)(kGuideUnroll - 1); };
    // pass 1: the wave's total (a lane adds up its column, the columns meet in the last lane of a scan)
    uint32_t column = 0;
#pragma unroll 16
    for (int i = 0; i < kRows; ++i) column += padded(c[i * kWave]);
    const uint32_t mine = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(column), kWave - 1);
    if (lane == 0) s_total[wave] = mine;
    block_sync();
    uint32_t run = 0;  // padded entries of all lists before row i of this wave
#pragma unroll
    for (int w = 0; w < kWaves; ++w) run += (uint32_t)w < wave ? s_total[w] : 0u;
    // pass 2: the rows again (from L2), eight in flight
    uint32_t *const p = poff + wave * kPerWave + lane;
    for (int i0 = 0; i0 < kRows; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = padded(c[(i0 + j) * kWave]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t inc = wave_inclusive_scan(v[j]);
            p[(i0 + j) * kWave] = run + inc - v[j];
            c[(i0 + j) * kWave] = 0;
            run += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
        }
    }
    if (threadIdx.x == 1023) poff[kLists] = run;
}

// count: kLists words of scratch; poff: kLists + 1 list starts (multiples of kGuideUnroll); list_rest: the lists,
// pre-filled with the padding pattern (y = ~0) by the caller
hipError_t launch_seed_lists(const uint2 *guides, uint32_t n_guides, const SeedPlan &plan, uint32_t *count, uint32_t *poff, uint2 *list_rest,
                             hipStream_t stream)
{
    const uint64_t total = (uint64_t)n_guides * kSegments * plan.n_nbr;
    hipError_t e = hipMemsetAsync(count, 0, (size_t)kLists * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (total) hipLaunchKernelGGL(seed_enum_kernel<false>, dim3(blocks), dim3(256), 0, stream, guides, n_guides, plan, count, (const uint32_t *)nullptr, (uint2 *)nullptr);
    hipLaunchKernelGGL(seed_pad_scan_kernel, dim3(1), dim3(1024), 0, stream, count, poff);
    if (total) hipLaunchKernelGGL(seed_enum_kernel<true>, dim3(blocks), dim3(256), 0, stream, guides, n_guides, plan, count, (const uint32_t *)poff, list_rest);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// bit-sliced comparison
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
typedef uint32_t v8u __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) v8u *const_v8u_ptr;  // a group of four 8-byte list entries, scalar-loaded

struct SeedWave {
    uint2 *tok;      // pending hit tokens (see sliced_fetch)
    uint32_t ntok, lane;
    uint32_t thead;  // ring slot of the oldest pending token
    uint32_t *parts; // per output region: this wave's open block, block number << 11 | records claimed in it
    uint32_t *first; // first site of every chunk of the current grab (tokens name their chunk by slot)
    uint32_t *info;  // ... and word z of its chunk table entry: bucket | rank of the first '-' site << 16 | edge flag << 28
};

// A per-pair comparison (xor, or, popcount, compare) spends 4 VALU instructions per (site, read) pair and lane.
// Here a lane holds 32 sites "vertically": one word per read position and plane, bit i = site i.  One VALU instruction then
// works on 32 pairs:  14 positions x 2 (mismatch vector) + a carry-save adder tree (10 full adders + 1 half adder, 2
// instructions each) + a 4-instruction bit-sliced "count <= budget" = 55 instructions per 32 pairs, no popcount, no
// per-pair branch.  Only the 14 positions of the OTHER TWO segments are compared: all sites of a bucket share the segment,
// whose distance d to the read is a property of the list entry (0, 1 or 2 substitutions), and all sites of a chunk share
// their PAM, so the budget for the rest is what the class leaves of m, less d (the entry carries it per class).
//
// Vertical block of 32 sites (28 words): words 0..13 = hi-plane bit of rest position q = 0..13, words 14..27 =
// lo-plane bit.  Block b of a chunk holds the chunk's sites [32 b, 32 b + 32); the blocks of a chunk are
// stored interleaved by word quad (see seed_transpose_kernel) so that a wave's loads are contiguous.
__device__ __forceinline__ uint32_t rest_position(uint32_t q, uint32_t seg)
{
    if (seg == 0) return q + kSegBases;
    if (seg == 1) return q < (uint32_t)kSegBases ? q : q + kSegBases;
    return q;  // (q < 14)
}

// One wave per chunk: 64 sites per step, two blocks; word j of a block is the ballot of one plane bit.
__global__ __launch_bounds__(kWave *kWavesPerGroup) void seed_transpose_kernel(const uint4 *sites, const uint4 *chunk_tab,
                                                                              uint32_t n_chunks, uint32_t *vert)
{
    const uint32_t c = blockIdx.x * kWavesPerGroup + threadIdx.x / kWave;
    if (c >= n_chunks) return;
    const uint32_t lane = threadIdx.x % kWave;
    const uint4 ct = chunk_tab[c];
    const uint32_t seg = (ct.z & kChunkBucketMask) / (uint32_t)kBucketsPerSeg;
    for (uint32_t k = 0; k * kWave < ct.y; ++k) {
        const uint32_t i = k * kWave + lane;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < ct.y) v = sites[ct.x + i];
        uint32_t mine = 0;
        for (uint32_t q = 0; q < (uint32_t)kCmpBases; ++q) {
            const uint32_t p = rest_position(q, seg);
            const uint64_t bh = __ballot((v.x >> p) & 1u), bl = __ballot((v.y >> p) & 1u);
            // lanes 0..31 assemble block 2k (low halves), lanes 32..63 block 2k + 1 (high halves)
            const uint32_t wh = lane < 32 ? (uint32_t)bh : (uint32_t)(bh >> 32);
            const uint32_t wl = lane < 32 ? (uint32_t)bl : (uint32_t)(bl >> 32);
            if ((lane & 31u) == q) mine = wh;
            if ((lane & 31u) == q + kCmpBases) mine = wl;
        }
        const uint32_t block = 2 * k + (lane >> 5), j = lane & 31u;
        const uint32_t nb = (ct.y + kSlicedSites - 1) / kSlicedSites;  // blocks of this chunk
        // chunk layout [word quad j / 4][block][j % 4]: the 16-byte loads of a wave (lane = block) coalesce
        if (block < nb && j < (uint32_t)kVertWords) vert[(size_t)ct.w * kVertWords + ((size_t)(j >> 2) * nb + block) * 4 + (j & 3u)] = mine;
    }
}

hipError_t launch_seed_transpose(const uint4 *sites, const uint4 *chunk_tab, uint32_t n_chunks, uint32_t *vert,
                                 hipStream_t stream)
{
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(seed_transpose_kernel, dim3((n_chunks + kWavesPerGroup - 1) / kWavesPerGroup),
                       dim3(kWave * kWavesPerGroup), 0, stream, sites, chunk_tab, n_chunks, vert);
    return hipGetLastError();
}

// v_bitop3_b32 with an explicit truth table: bit (a << 2 | b << 1 | c) of `kTable` is the result for the
// input bits (a, b, c), i.e. kTable = f(0xF0, 0xCC, 0xAA).  Written out because the compiler, given the
// boolean expressions, shares sub-terms between sum and carry and ends up with three instructions per
// full adder instead of two.
template <int kTable> __device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, kTable);
}

__device__ __forceinline__ void full_add(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum, uint32_t &carry)
{
    sum = bitop3<0x96>(a, b, c);    // a ^ b ^ c
    carry = bitop3<0xE8>(a, b, c);  // majority
}

__device__ __forceinline__ void half_add(uint32_t a, uint32_t b, uint32_t &sum, uint32_t &carry)
{
    sum = a ^ b;
    carry = a & b;
}

// all-ones if bit `bit` of the (wave-uniform) word is set
__device__ __forceinline__ uint32_t spread(uint32_t word, int bit)
{
    return (uint32_t)((int32_t)(word << (31 - bit)) >> 31);
}

// 7 one-bit inputs -> 3-bit count (4 full adders)
__device__ __forceinline__ void count7(const uint32_t *m, uint32_t &b0, uint32_t &b1, uint32_t &b2)
{
    uint32_t s1, c1, s2, c2, c3;
    full_add(m[0], m[1], m[2], s1, c1);
    full_add(m[3], m[4], m[5], s2, c2);
    full_add(s1, s2, m[6], b0, c3);
    full_add(c1, c2, c3, b1, b2);
}

// Bit i of the result: site i of this lane's block is valid, within `budget` mismatches of the read on the
// 14 compared positions, and NOT already reported by an earlier segment.  rx = rest(hi) | rest(lo) << 16 of the
// read (wave-uniform), budget <= 8 (what the chunk's class and the seed distance leave of max_mm: seed_enum_kernel).
// Whatever the bucket's segment, the compared positions are [7 positions of another segment][7 of the third], so the adder
// tree counts the two groups of seven (3 bits each) and adds them: the group counts give the duplicate test for free - a
// pair with <= k mismatches in an EARLIER segment (group A for segments 1 and 2, also group B for segment 2) was reported
// from that segment's bucket.  Without this 40 % of the candidates at m = 8 (1.67 qualifying segments per hit on average)
// went through the hit path only to be dropped there.
// History: 72 instructions with read positions 21 and 22 in the comparison (5-bit count), 57 + 3 once the index's common
// first PAM letter had left it (round 3), 55 + 3 now that the chunk's class settles both PAM letters.
template <uint32_t kSeg>
__device__ __forceinline__ uint32_t sliced_within(const uint32_t (&v)[kVertWords], uint32_t rx, uint32_t budget, uint32_t valid,
                                                  const uint32_t (&kv)[4])
{
    uint32_t mm[kCmpBases];
#pragma unroll
    for (int q = 0; q < kCmpBases; ++q)  // (hi ^ read hi) | (lo ^ read lo)
        mm[q] = bitop3<0xF6>(v[q] ^ spread(rx, q), v[kCmpBases + q], spread(rx, kRestBases + q));
    uint32_t a0, a1, a2, b0, b1, b2;
    count7(mm, a0, a1, a2);
    count7(mm + kSegBases, b0, b1, b2);
    // A + B: a 4-bit count
    uint32_t t0, t1, t2, k0, k1, k2;
    half_add(a0, b0, t0, k0);
    full_add(a1, b1, k0, t1, k1);
    full_add(a2, b2, k1, t2, k2);
    // count <= budget, from the least significant bit up: le_i = (~c_i & b_i) | (~(c_i ^ b_i) & le_{i-1})
    uint32_t le = ~t0 | spread(budget, 0);
    le = bitop3<0x8E>(t1, spread(budget, 1), le);
    le = bitop3<0x8E>(t2, spread(budget, 2), le);
    le = bitop3<0x8E>(k2, spread(budget, 3), le);
    uint32_t ok = le & valid;
    // Not reported by an earlier segment's bucket: its group count > that segment's threshold.  A threshold is <= 2, so
    // bit 2 of the count alone says "greater"; kv = bits 0 and 1 of the thresholds of segments 0 (kv[0], kv[1]) and 1 (kv[2],
    // kv[3]), spread, in VECTOR registers: an instruction
    // with a scalar operand issues at 4.2 cycles per SIMD, with vector operands only at 2.6 (tools/micro/valu_rate.hip).
    if (kSeg >= 1) ok = bitop3<0xD0>(ok, a2, bitop3<0x8E>(a1, kv[1], ~a0 | kv[0]));  // ok & (a2 | ~le)
    if (kSeg >= 2) ok = bitop3<0xD0>(ok, b2, bitop3<0x8E>(b1, kv[3], ~b0 | kv[2]));  // (segment 2: group B = segment 1)
    return ok;
}

// Second half of the hit path.  A token = {hit word of one lane's block, read | chunk slot in the grab << 14 |
// owner lane << 26} (8 bytes: a ring of seven passes is 3.5 KB per wave, which is what lets five waves per SIMD
// share the CU's LDS); a lane takes one token and resolves its LOWEST set bit: one 8-byte gather of the site
// record (rest planes, position), one of the read's planes (a table of 8 bytes per read: cache resident) + the
// bucket's code and the chunk's strand boundary (LDS) give the full 23-position mask, the strand and the position.  A token with more bits goes
// back into the ring with that bit cleared, so every pass over 64 tokens is dense.  The gather of the next
// 64 tokens is issued before the current 64 are consumed.
struct SlicedFetch {
    uint32_t word, hi, site0;  // hi = read | chunk slot << 14 | owner lane << 26, as in the token
    uint2 gp;   // read planes
    uint2 rec;  // site record of the lowest set bit
};

// Output of the sliced kernel.  A wave keeps, per output region (64 reads), an open block of `reserve`
// (a power of two, <= 1024) reserved records, one 32-bit word of LDS per region: block number << 11 | records
// claimed in it (the block starts at record block number x reserve of the region).
// The hits of a pass belong to whatever regions their reads fall into - with 128.8 reads per bucket spread over
// 157 regions (c3) that is five or six regions per pass - so every hit lane claims its slot by itself with ONE
// LDS atomic on its region's word and stores straight from registers; nothing in the common path depends on
// how many regions a pass touches.  Only when a block is used up (once per `reserve` hits of a region) do the
// lanes that found it full take the wave-uniform path that reserves the next block.
// State word of an open block: records claimed (11 bits) | BUSY (group-shared blocks: a wave is reserving the next
// block) | block number (20 bits; all ones: the region is full).
constexpr uint32_t kPartUsedBits = 11;
constexpr uint32_t kPartUsedMask = (1u << kPartUsedBits) - 1u;
constexpr uint32_t kPartBusy = 1u << kPartUsedBits;
constexpr uint32_t kPartBlockShift = kPartUsedBits + 1;
constexpr uint32_t kPartDeadBlock = 0xFFFFFu;

// Reserves a new block in region p (wave-uniform).  Returns false when the region is full: the host
// re-runs with room for reserved + lost records in every region.
__device__ __forceinline__ bool sliced_reserve(const SeedArgs &a, const SeedWave &w, uint32_t p, uint32_t losing, uint32_t &base)
{
    unsigned long long *const cursor = &a.counters[kCntPart + 4 * p];
    unsigned long long rel = 0;
    if (w.lane == 0) rel = atomicAdd(cursor, (unsigned long long)a.reserve);
    rel = ((unsigned long long)uniform((uint32_t)(rel >> 32)) << 32) | uniform((uint32_t)rel);
    if (rel + a.reserve > a.part_cap) {
        if (w.lane == 0) {
            atomicAdd(cursor, 0ull - (unsigned long long)a.reserve);
            atomicAdd(cursor + 2, (unsigned long long)losing);
            atomicMax(&a.counters[kCntOverflow], 1ull);
        }
        return false;
    }
    base = (uint32_t)rel;  // part_cap < 2^32; a multiple of `reserve` (every reservation is)
    return true;
}

// record slot `at` of the output regions (+ the parallel side word: a search that keeps the sites' bases - SeedArgs.hit_side -
// writes the site's lo plane there, its hi plane sits in the record where the mismatch mask is otherwise)
__device__ __forceinline__ void put_record(const SeedArgs &a, unsigned long long at, uint64_t rec, uint32_t side)
{
    a.hit_recs[at] = rec;
    if (a.hit_side) a.hit_side[at] = side;
}

// Group-shared open blocks (SeedArgs.group_out; with chunk sharing only): the four waves of a workgroup fill ONE open
// block per region - a quarter of the partly written lines in flight and of the sentinels at the end, blocks four times
// as large for the same padding (fewer reservations).  The slow path serialises per region on the BUSY bit: the wave
// that finds the block full and not busy sets BUSY (LDS compare-and-swap), reserves the next block (global atomic) and
// installs it with a plain LDS store that clears BUSY; a wave that finds BUSY waits (s_sleep, bounded) and then claims
// again.  A wave holds BUSY only across its own reservation, so nobody waits on a waiter.
__device__ __forceinline__ void sliced_store_group_slow(const SeedArgs &a, SeedWave &w, uint32_t p, bool mine, uint64_t rec, uint32_t side)
{
    volatile uint32_t *const word = &w.parts[p];
    for (uint32_t spins = 0;;) {
        uint32_t s = 0;
        if (w.lane == 0) s = *word;
        s = uniform(s);
        const uint32_t used = s & kPartUsedMask;
        if ((s >> kPartBlockShift) == kPartDeadBlock) {
            // the region is full: these hits are lost, the host re-runs with more room (and the claim counter starts again)
            const uint64_t b = __ballot(mine);
            if (w.lane == 0) {
                atomicAdd(&a.counters[kCntPart + 4 * p + 2], (unsigned long long)__popcll(b));
                *word = (kPartDeadBlock << kPartBlockShift) | a.reserve;
            }
            return;
        }
        if (!(s & kPartBusy) && used < a.reserve) {
            // somebody installed a block with room: claim again
            uint32_t st = kPartUsedMask;
            if (mine) st = atomicAdd(&w.parts[p], 1u);
            const bool ok = mine && (st & kPartUsedMask) < a.reserve;
            if (ok) put_record(a, (unsigned long long)p * (uint32_t)a.part_cap + (((st >> kPartBlockShift) << a.reserve_log2) + (st & kPartUsedMask)), rec, side);
            mine = mine && !ok;
            if (__ballot(mine) == 0) return;
            continue;
        }
        if (s & kPartBusy) {
            if (++spins > (1u << 20)) {  // (never seen; a safety net instead of a hang: counted as lost, the host re-runs)
                const uint64_t b = __ballot(mine);
                if (w.lane == 0) {
                    atomicAdd(&a.counters[kCntPart + 4 * p + 2], (unsigned long long)__popcll(b));
                    atomicMax(&a.counters[kCntOverflow], 1ull);
                }
                return;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        // full and nobody is reserving: this wave does, if it wins the compare-and-swap
        uint32_t won = 0;
        if (w.lane == 0) won = atomicCAS(&w.parts[p], s, s | kPartBusy) == s ? 1u : 0u;
        if (!uniform(won)) continue;
        const uint64_t b = __ballot(mine);
        const uint32_t n = (uint32_t)__popcll(b);
        uint32_t next = 0;
        const bool ok = sliced_reserve(a, w, p, n, next);
        if (mine && ok) put_record(a, (unsigned long long)p * a.part_cap + next + lanes_below(b), rec, side);
        if (w.lane == 0) *word = ok ? ((next >> a.reserve_log2) << kPartBlockShift) | n : (kPartDeadBlock << kPartBlockShift) | a.reserve;
        return;
    }
}

// lanes with `hit` store their record (+ its side word, when the search keeps the sites' bases) in the region of their read
__device__ __forceinline__ void sliced_store(const SeedArgs &a, SeedWave &w, bool hit, uint32_t region, uint64_t rec, uint32_t side)
{
    uint32_t state = kPartUsedMask;
    if (hit) state = atomicAdd(&w.parts[region], 1u);  // LDS; the count may run past `reserve`: those lanes take the path below
    const uint32_t slot = state & kPartUsedMask;
    const bool placed = hit && slot < a.reserve;
    // (part_cap < 2^32 - the host checks - so the slot index is one 32 x 32 -> 64-bit multiply-add)
    if (placed) put_record(a, (unsigned long long)region * (uint32_t)a.part_cap + (((state >> kPartBlockShift) << a.reserve_log2) + slot), rec, side);
    uint64_t todo = __ballot(hit && !placed);
    while (todo != 0) {
        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)region, (int)__builtin_ctzll(todo));
        const bool mine = hit && !placed && region == p;
        const uint64_t b = __ballot(mine);
        if (a.group_out) {
            sliced_store_group_slow(a, w, p, mine, rec, side);
        } else {
            const uint32_t n = (uint32_t)__popcll(b);
            uint32_t next = 0;
            const bool ok = sliced_reserve(a, w, p, n, next);
            if (mine && ok) put_record(a, (unsigned long long)p * a.part_cap + next + lanes_below(b), rec, side);
            // (a region that is full keeps a "used up" block: later hits come here again and are counted as lost)
            if (w.lane == 0) w.parts[p] = ok ? ((next >> a.reserve_log2) << kPartBlockShift) | n : a.reserve;
        }
        wave_sync();
        todo &= ~b;
    }
}

// end of the kernel: the open blocks are filled up with sentinels, which the sort drops (group-shared blocks: behind
// the workgroup's last barrier, region q by wave q mod 4)
__device__ __forceinline__ void sliced_finish_hits(const SeedArgs &a, SeedWave &w, uint32_t first, uint32_t step)
{
    wave_sync();
    for (uint32_t q = first; q < a.n_parts; q += step) {
        const uint32_t state = uniform(w.parts[q]);
        if ((state >> kPartBlockShift) == kPartDeadBlock) continue;
        const uint32_t base = (state >> kPartBlockShift) << a.reserve_log2, used = state & kPartUsedMask;
        const uint32_t left = used < a.reserve ? a.reserve - used : 0u;
        for (uint32_t i = w.lane; i < left; i += kWave) a.hit_recs[(unsigned long long)q * a.part_cap + base + used + i] = kRecSentinel;
        if (w.lane == 0 && left) atomicAdd(&a.counters[kCntPart + 4 * q + 1], (unsigned long long)left);
    }
}

// ring slot of running position p
__device__ __forceinline__ uint32_t ring_slot(uint32_t p)
{
    return p & (uint32_t)(kSlicedTokCap - 1);
}

// ring slots head .. head + n - 1 -> one token per lane (lanes >= n stay empty), gather issued
// (kFull: n == 64, every lane takes a token - no predication, nothing to clear)
template <bool kFull>
__device__ __forceinline__ SlicedFetch sliced_fetch(const SeedArgs &a, const SeedWave &w, uint32_t head, uint32_t n)
{
    SlicedFetch f;
    if (!kFull) {
        f.word = 0;
        f.hi = 0;
        f.site0 = 0;
        f.gp = make_uint2(0u, 0u);
        f.rec = make_uint2(0u, 0u);
    }
    if (kFull || w.lane < n) {
        const uint2 tk = w.tok[ring_slot(head + w.lane)];
        f.word = tk.x;
        f.hi = tk.y;
        f.gp = a.guides[tk.y & kTokReadMask];
        // the token's chunk is one of the kSlicedGrab chunks of the current grab (w.first = their first sites)
        f.site0 = w.first[(tk.y >> kTokSlotShift) & (uint32_t)(kSlicedGrab - 1)] + (tk.y >> kTokLaneShift) * kSlicedSites;
        f.rec = a.sites[f.site0 + (uint32_t)__builtin_ctz(tk.x)];
    }
    return f;
}

// w.ntok tokens wait in the ring from slot w.thead on; `tail` = first free slot
__device__ __forceinline__ uint32_t ring_tail(const SeedWave &w) { return ring_slot(w.thead + w.ntok); }

template <bool kFull>
__device__ __forceinline__ void sliced_consume(const SeedArgs &a, SeedWave &w, const SlicedFetch &f)
{
    bool hit = kFull || f.word != 0;
    // more hits of the same (block, read): back into the ring
    const uint32_t rest = f.word & (f.word - 1);
    const uint64_t again = __ballot(rest != 0);
    if (again != 0) {
        if (rest != 0) w.tok[ring_slot(lanes_below(again, ring_tail(w)))] = make_uint2(rest, f.hi);
        w.ntok += (uint32_t)__popcll(again);
    }
    // Straight-line, computed for every lane (a lane without a token works on zeros): branches around the
    // few instructions cost more scalar mask bookkeeping than the instructions themselves.  Only the
    // right-edge rule - a binary search, needed by windows that end a contig - sits behind a wave-uniform test.
    const uint32_t gid_of = f.hi & kTokReadMask;
    // the site's 23-base planes: its 16 rest positions from the record, the 7 segment positions = the bucket's code
    const uint32_t info = w.info[(f.hi >> kTokSlotShift) & (uint32_t)(kSlicedGrab - 1)];  // chunk table word z of the token's chunk
    const uint32_t bucket = info & kChunkBucketMask;
    const uint32_t sh = (uint32_t)kSegBases * (bucket >> (2 * kSegBases));  // 7 x segment
    const uint32_t low = (1u << sh) - 1u;
    const uint32_t rest_hi = f.rec.x & 0xFFFFu, rest_lo = f.rec.x >> 16;
    const uint32_t site_hi = (rest_hi & low) | (((bucket >> kSegBases) & 0x7Fu) << sh) | ((rest_hi >> sh) << (sh + kSegBases));
    const uint32_t site_lo = (rest_lo & low) | ((bucket & 0x7Fu) << sh) | ((rest_lo >> sh) << (sh + kSegBases));
    const uint32_t t = ((site_hi ^ f.gp.x) | (site_lo ^ f.gp.y)) & kMask23;
    // '-' sites are the tail of their bucket: from rank `minus_from` of the chunk on
    const uint32_t rank_in_chunk = (f.hi >> kTokLaneShift) * kSlicedSites + (uint32_t)__builtin_ctz(f.word | 0x80000000u);
    const uint32_t strand = rank_in_chunk >= ((info >> kChunkMinusShift) & 0xFFFu) ? 1u : 0u;
    const uint32_t pos = f.rec.y;
    // (pairs that an earlier segment reports never become tokens: sliced_within)
    const uint32_t mask = strand ? reverse23(t) : t;
    // right-edge rule, bidir_mapping.cpp:51-52 (see emit_hits in vsc_kernels.hip): only chunks that hold a window
    // followed by N are flagged, and only their hits look the site up in the bitmap
    const bool maybe_edge = hit && ((info >> kChunkEdgeBit) & 1u) && (uint32_t)__popc(mask >> (VSC_READ_LEN / 2)) > a.k_half;
    if (__ballot(maybe_edge) != 0) {
        if (maybe_edge) {
            const uint32_t site = f.site0 + (uint32_t)__builtin_ctz(f.word);
            if (((a.edge_bits[site >> 5] >> (site & 31u)) & 1u) && is_contig_end(a.contig_end, a.n_contigs, pos + VSC_READ_LEN)) hit = false;
        }
    }
    // (a search that keeps the sites' bases for the per-hit feature rows: the site's planes in READ orientation - hi in the
    // record's low 23 bits, lo in the side word - instead of the mask, which the record assembly recomputes from them)
    const uint32_t low23 = a.hit_side ? (site_hi & kMask23) : mask;
    const uint64_t rec = ((uint64_t)(gid_of & (uint32_t)(kRegionReads - 1)) << kRecReadShift) | ((uint64_t)strand << kRecStrandShift) |
                         ((uint64_t)((pos - a.pos_base) << a.pos_pad) << kRecPosShift) | low23;
    sliced_store(a, w, hit, gid_of >> kRegionBits, rec, site_lo & kMask23);
}

// Resolves tokens in passes of 64.  kDrain = false: full passes only - what is left (< 64 tokens) waits
// for more, so that every pass is dense (and its code free of predication); kDrain = true (end of a grab of
// chunks): everything.
template <bool kDrain>
__device__ __forceinline__ void sliced_resolve(const SeedArgs &a, SeedWave &w)
{
    wave_sync();
    bool have = false;
    SlicedFetch f = sliced_fetch<false>(a, w, 0, 0);
    for (;;) {
        uint32_t n = min(w.ntok, (uint32_t)kWave);
        if (!kDrain && n < (uint32_t)kWave) n = 0;
        if (n == 0 && !have) break;
        SlicedFetch nf = f;  // n == 0: not looked at again
        if (kDrain) nf = sliced_fetch<false>(a, w, w.thead, n);
        else if (n) nf = sliced_fetch<true>(a, w, w.thead, n);
        w.thead = ring_slot(w.thead + n);
        w.ntok -= n;
        if (have) sliced_consume<!kDrain>(a, w, f);  // may append to the ring
        wave_sync();
        f = nf;
        have = n != 0;
    }
    // (wave-uniform, but computed on the vector unit in here: handed back as scalars, or the comparison loop's token
    // bookkeeping - three instructions per read - stays on the vector unit as well)
    w.ntok = uniform(w.ntok);
    w.thead = uniform(w.thead);
}

// entry g0 + lane of a read list [g0, g1) (past the end: padding, y = ~0): rest planes, read | budget per class
__device__ __forceinline__ uint2 sliced_load_list(const SeedArgs &a, uint32_t g0, uint32_t g1, uint32_t lane)
{
    uint2 e = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    if (g0 + lane < g1) e = a.list_rest[g0 + lane];
    return e;
}

// this lane's block of the chunk's bit-sliced sites (lanes past the chunk's last block: zeros)
__device__ __forceinline__ void sliced_load_sites(const SeedArgs &a, const v4u &ct, bool wanted, uint32_t lane,
                                                  uint32_t (&v)[kVertWords])
{
    const uint32_t nb = (ct.y + kSlicedSites - 1) / kSlicedSites;
    const bool mine = wanted && lane < nb;
    const uint4 *vp = (const uint4 *)(a.vert + (size_t)ct.w * kVertWords) + lane;
#pragma unroll
    for (int j = 0; j < kVertWords / 4; ++j) {
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if (mine) x = vp[(size_t)j * nb];
        v[4 * j] = x.x;
        v[4 * j + 1] = x.y;
        v[4 * j + 2] = x.z;
        v[4 * j + 3] = x.w;
    }
}


// (amdgpu_waves_per_eu: 5 waves per SIMD = at most 102 VGPRs; LDS: 5.7 KB per wave, 20 waves per CU)
static_assert(kMaxPassReads <= (1 << kTokSlotShift) && kSlicedGrab <= 8 && kTokSlotShift + 3 <= kTokLaneShift, "token fields");
static_assert((kSlicedTokCap & (kSlicedTokCap - 1)) == 0 && kSlicedTokCap >= (kSlicedResolve + kGuideUnroll) * kWave, "token ring");
// kShared: the four waves of a workgroup take the SAME chunks and a quarter of each chunk's read list each.  The
// resident site records (16 KB per chunk, gathered 8 bytes per hit) then are a working set of one chunk per
// workgroup instead of one per wave - 3 MB per XCD instead of 12 - which is what its 4 MB L2 can hold: with a chunk
// per wave every gather of a hit fetched its line from the fabric again (FETCH_SIZE 65 GB per c3 search; 17 GB with
// one workgroup per CU resident, tools/experiments.sh groups).  For sparse searches (c2: 13 reads per bucket) a chunk visit
// is mostly the load of its bit-sliced block, which every sharing wave repeats: those keep a chunk per wave.
template <bool kShared>
__global__ __launch_bounds__(kWave *kWavesPerGroup) __attribute__((amdgpu_waves_per_eu(kSlicedWavesPerSimd, kSlicedWavesPerSimd))) void seed_sliced_kernel(
    const SeedArgs a)
{
    __shared__ uint32_t s_grab[2];
    __shared__ uint2 s_tok[kWavesPerGroup][kSlicedTokCap];
    __shared__ uint2 s_list[kWavesPerGroup][kWave];  // the current tile of 64 read-list entries (first halves)
    __shared__ uint32_t s_parts[kWavesPerGroup][kParts];
    __shared__ uint32_t s_first[kWavesPerGroup][2 * kSlicedGrab];

    const uint32_t wave = threadIdx.x / kWave;
    SeedWave w;
    w.lane = threadIdx.x % kWave;
    w.tok = s_tok[wave];
    w.ntok = 0;
    w.thead = 0;
    w.parts = s_parts[kShared && a.group_out ? 0u : wave];  // (group-shared blocks: first used behind the first grab's barrier)
    w.first = s_first[wave];
    w.info = s_first[wave] + kSlicedGrab;
    for (uint32_t q = w.lane; q < (uint32_t)kParts; q += kWave) w.parts[q] = a.reserve;  // no block yet = a used-up one

    uint2 *const lt = s_list[wave];
    uint32_t kv[4];  // bits 0 and 1 of the thresholds of segments 0 and 1, spread (the duplicate test of sliced_within)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(kv[i]) : "s"(spread(i < 2 ? a.k_seg0 : a.k_seg1, i & 1)));
    const const_v4u_ptr ctab = (const_v4u_ptr)(uintptr_t)a.chunk_tab;
    const const_u32_ptr poff = (const_u32_ptr)(uintptr_t)a.poff;
    const const_v8u_ptr lgrp = (const_v8u_ptr)(uintptr_t)a.list_rest;
    constexpr bool kScalarLists = kShared;
    // the read list of a chunk (word z of its table entry): its bucket's, for segment 2 its class's list of that bucket
    auto list_of = [](uint32_t z) {
        const uint32_t bucket = z & kChunkBucketMask;
        return bucket + (bucket >= 2u * kBucketsPerSeg ? ((z >> kChunkClassShift) & 3u) * (uint32_t)kBucketsPerSeg : 0u);
    };
    const uint32_t lane_tag = w.lane << kTokLaneShift;
    unsigned long long pairs = 0, visited = 0;

    // work distribution: the chunk range is cut into kCursors slices with a cursor each; a wave starts in
    // "its" slice and moves on to the next one when a slice is used up (see kCursors)
    const uint32_t slice_len = (a.n_chunks + kCursors - 1) / kCursors;
    // (uniform(): `wave` derives from threadIdx, and a slice the compiler takes for divergent drags the whole
    // chunk bookkeeping - table entries, list bounds, loop counters - from scalar into vector registers)
    uint32_t slice = uniform((kShared ? blockIdx.x : blockIdx.x * kWavesPerGroup + wave) % kCursors), exhausted = 0;
    // the rank of this wave among the four that share a chunk rotates from chunk to chunk: the quarters of a list
    // differ by up to four entries, and consecutive chunks mostly belong to the same bucket
    const uint32_t wave_u = uniform(wave);
    for (uint32_t iter = 0;; ++iter) {
        uint32_t off = 0;
        if (kShared) {
            // one grab per workgroup: wave 0 asks, the others read its answer behind the barrier.  Every wave sees the
            // same sequence of answers, so `slice` and `exhausted` evolve identically and all four leave together.
            // (Two slots: a wave may be one iteration ahead of the slowest reader, never two - the barrier.)
            if (threadIdx.x == 0)
                s_grab[iter & 1u] = (uint32_t)atomicAdd(&a.counters[kCursorBase + slice * kCursorStride], (unsigned long long)kSlicedGrab);
            block_sync();
            off = s_grab[iter & 1u];
        } else if (w.lane == 0) {
            off = (uint32_t)atomicAdd(&a.counters[kCursorBase + slice * kCursorStride], (unsigned long long)kSlicedGrab);
        }
        off = uniform(off);
        const uint32_t slice_begin = slice * slice_len, slice_end = min(slice_begin + slice_len, a.n_chunks);
        if (slice_begin + off >= slice_end) {
            if (++exhausted == (uint32_t)kCursors) break;
            slice = (slice + 1) % kCursors;
            continue;
        }
        const uint32_t first = slice_begin + off;
        const uint32_t last = min(first + (uint32_t)kSlicedGrab, slice_end);
        // Software pipeline over the chunks of this grab: the chunk table entry (scalar load) runs two chunks ahead,
        // the read-list bounds poff[bucket..] (scalar load) and the first 64 entries of the list one chunk ahead.
        // The lane's block of bit-sliced sites (8 x 16-byte vector loads) is NOT fetched ahead: holding the next
        // block cost 32 VGPRs, and at 93 instead of 125 a fifth wave per SIMD fits, which covers more than the one
        // exposed round trip per chunk (c3: 26.4 -> 26.0 ms, c2: 2.28 -> 2.07 ms; without the fifth wave 26.6 / 2.27).
        // this wave's part [q0, q1) of the read list [g0, g1) of a chunk: all of it, or (kShared) the quarter of rank r -
        // whole groups of kGuideUnroll entries
        auto my_part = [&](uint32_t g0, uint32_t g1, uint32_t r, uint32_t &q0, uint32_t &q1) {
            if (!kShared) {
                q0 = g0;
                q1 = g1;
                return;
            }
            const uint32_t n4 = (g1 - g0) / (uint32_t)kGuideUnroll;
            const uint32_t base = n4 / (uint32_t)kWavesPerGroup, rem = n4 % (uint32_t)kWavesPerGroup;
            q0 = g0 + (r * base + min(r, rem)) * (uint32_t)kGuideUnroll;
            q1 = q0 + (base + (r < rem ? 1u : 0u)) * (uint32_t)kGuideUnroll;
        };
        v4u t0 = ctab[first];                             // chunk c
        v4u t1 = ctab[min(first + 1, last - 1)];          // chunk c + 1
        uint32_t p0a, p0b;
        my_part(poff[list_of(t0.z)], poff[list_of(t0.z) + 1], wave_u, p0a, p0b);
        uint2 nl = make_uint2(0u, 0u);
        v8u ng = {};  // kScalarLists: the first group of the next chunk's part
        if (kScalarLists) ng = lgrp[(p0a < p0b ? p0a : 0u) / (uint32_t)kGuideUnroll];
        else nl = sliced_load_list(a, p0a, p0b, w.lane);
        uint32_t p1a = poff[list_of(t1.z)], p1b = poff[list_of(t1.z) + 1];
        v4u t2 = ctab[min(first + 2, last - 1)];          // chunk c + 2
        for (uint32_t c = first; c < last; ++c) {
            const v4u cur = t0;
            const uint32_t g0 = p0a, g1 = p0b;
            uint32_t v[kVertWords];
            sliced_load_sites(a, cur, g0 != g1, w.lane, v);
            // advance the pipeline before the comparison so that its loads overlap it
            t0 = t1;
            my_part(p1a, p1b, (wave_u + (c + 1 - first)) % (uint32_t)kWavesPerGroup, p0a, p0b);
            uint2 tile = nl;
            v8u grp = ng;
            if (kScalarLists) ng = lgrp[(p0a < p0b && c + 1 < last ? p0a : 0u) / (uint32_t)kGuideUnroll];
            else nl = sliced_load_list(a, p0a, c + 1 < last ? p0b : p0a, w.lane);
            t1 = t2;
            p1a = poff[list_of(t1.z)];
            p1b = poff[list_of(t1.z) + 1];
            t2 = ctab[min(c + 3, last - 1)];
            if (g0 == g1) continue;  // no read has this bucket in its neighbourhood
            const uint32_t slot_tag = (c - first) << kTokSlotShift;  // the tokens of this chunk carry its slot in the grab
            if (w.lane == 0) {
                w.first[c - first] = cur.x;
                w.info[c - first] = cur.z;
            }
            wave_sync();
            // (uniform(): taken for divergent, the choice between the segments' copies of the comparison becomes three masked passes)
            const uint32_t seg = uniform((cur.z & kChunkBucketMask) / (uint32_t)kBucketsPerSeg);
            // where an entry keeps the budget of this chunk's class: s_bfe_u32's operand (offset | width << 16)
            const uint32_t budget_field = uniform((kListBudgetShift + 4u * ((cur.z >> kChunkClassShift) & 3u)) | (4u << 16));
            // sites of this lane's block that exist
            const int32_t left = (int32_t)cur.y - (int32_t)(w.lane * kSlicedSites);
            const uint32_t valid = left >= kSlicedSites ? 0xFFFFFFFFu : (left > 0 ? (1u << left) - 1u : 0u);
            pairs += (unsigned long long)cur.y * (g1 - g0);
            // (the block of a shared chunk comes from HBM once: counted by the wave of rank 0)
            if (!kShared || (wave_u + (c - first)) % (uint32_t)kWavesPerGroup == 0) visited += cur.y;
            // the four comparisons of a group of list entries (ex, ey: wave-uniform).  The duplicate test differs by segment - none,
            // group A, groups A and B; left as a run-time `if` inside the comparison the compiler computes all of it and
            // selects: one copy of the four comparisons per segment
            auto compare_four = [&](const uint32_t (&ex)[kGuideUnroll], const uint32_t (&ey)[kGuideUnroll]) {
                // first free ring slot and tokens added, as scalars for the length of the group (s_add by hand: left to the
                // compiler the wave's token counters live in vector registers - the resolve computes them there - and every
                // read pays three vector instructions for their bookkeeping)
                const uint32_t tail0 = uniform(w.thead + w.ntok);
                uint32_t tail = tail0;
                auto compare_group = [&](auto segment) {
                    constexpr uint32_t kSeg = decltype(segment)::value;
#pragma unroll
                    for (int u = 0; u < kGuideUnroll; ++u) {
                        const uint32_t ry = ey[u];
                        // what is left of max_mm for the compared positions against this chunk's class (seed_enum_kernel)
                        uint32_t budget;
                        asm("s_bfe_u32 %0, %1, %2" : "=s"(budget) : "s"(ry), "s"(budget_field) : "scc");
                        if (budget == kListNoBudget) continue;  // list padding, or a read that cannot reach this class
                        const uint32_t word = sliced_within<kSeg>(v, ex[u], budget, valid, kv);
                        const uint64_t b = __ballot(word != 0);
                        if (b == 0) continue;
                        const uint32_t gid = ry & kTokReadMask;
                        if (word != 0) w.tok[ring_slot(lanes_below(b, tail))] = make_uint2(word, gid | slot_tag | lane_tag);
                        const uint32_t cnt = (uint32_t)__popcll(b);
                        const uint32_t tail_was = tail;  // (asm operands cannot name the captures themselves)
                        uint32_t tail_now;
                        asm("s_add_i32 %0, %1, %2" : "=s"(tail_now) : "s"(tail_was), "s"(cnt) : "scc");
                        tail = tail_now;
                    }
                };
                if (seg == 0) compare_group(std::integral_constant<uint32_t, 0>{});
                else if (seg == 1) compare_group(std::integral_constant<uint32_t, 1>{});
                else compare_group(std::integral_constant<uint32_t, 2>{});
                w.ntok += tail - tail0;
                // a group of four reads adds at most 4 x 64 tokens, a resolve leaves fewer than 64
                if (w.ntok >= (uint32_t)kSlicedResolve * kWave) sliced_resolve<false>(a, w);
            };
            if (kScalarLists) {
                // Dense searches (the chunk-sharing kernel: a wave's part of a list is ~30 entries): the entries come as scalar
                // loads of a whole group (32 bytes), the next group requested before the current one is compared - no LDS
                // tile, no broadcast read, no v_readfirstlane per entry (two vector instructions per read in a kernel that
                // is bound by them).  The first group of a chunk's part was requested a chunk ago.
                for (uint32_t gi = g0; gi < g1; gi += kGuideUnroll) {
                    const v8u e = grp;
                    grp = lgrp[(gi + kGuideUnroll < g1 ? gi + kGuideUnroll : gi) / (uint32_t)kGuideUnroll];
                    const uint32_t ex[kGuideUnroll] = {e[0], e[2], e[4], e[6]}, ey[kGuideUnroll] = {e[1], e[3], e[5], e[7]};
                    compare_four(ex, ey);
                }
                continue;
            }
            // Sparse searches: the read list goes through LDS in tiles of 64 entries: one coalesced vector load per tile,
            // issued a whole tile (or chunk) ahead, then one broadcast LDS read per entry.  Scalar loads of
            // the entries, a group at a time, left ~1 us of scalar-cache miss latency per group exposed
            // whenever a bucket has only a handful of reads.
            for (uint32_t tg = g0; tg < g1; tg += kWave) {
                wave_sync();
                lt[w.lane] = tile;
                wave_sync();
                if (tg + kWave < g1) tile = sliced_load_list(a, tg + kWave, g1, w.lane);
                const uint32_t in_tile = min(g1 - tg, (uint32_t)kWave);
                for (uint32_t gi = 0; gi < in_tile; gi += kGuideUnroll) {
                    uint2 rd[kGuideUnroll];
#pragma unroll
                    for (int u = 0; u < kGuideUnroll; ++u) rd[u] = lt[gi + u];  // same address in every lane
                    uint32_t ex[kGuideUnroll], ey[kGuideUnroll];
#pragma unroll
                    for (int u = 0; u < kGuideUnroll; ++u) {
                        ex[u] = uniform(rd[u].x);
                        ey[u] = uniform(rd[u].y);
                    }
                    compare_four(ex, ey);
                }
            }
        }
        if (w.ntok) sliced_resolve<true>(a, w);  // the tokens name chunks of this grab: all out before the next
    }
    if (kShared && a.group_out) {
        block_sync();  // (the four waves leave the loop together: every store into the shared blocks is issued)
        sliced_finish_hits(a, w, wave_u, kWavesPerGroup);
    } else {
        sliced_finish_hits(a, w, 0, 1);
    }
    if (w.lane == 0 && pairs) {
        atomicAdd(&a.counters[kCntSites], pairs);
        atomicAdd(&a.counters[kCntVisited], visited);
    }
}

hipError_t launch_seed_sliced(const SeedArgs &args, int n_groups, bool shared, hipStream_t stream)
{
    const dim3 grid(n_groups), block(kWave * kWavesPerGroup);
    if (shared) hipLaunchKernelGGL((seed_sliced_kernel<true>), grid, block, 0, stream, args);
    else hipLaunchKernelGGL((seed_sliced_kernel<false>), grid, block, 0, stream, args);
    return hipGetLastError();
}

}  // namespace vsc
