// vsc_kernels.hip - CDNA4 (gfx950) kernels of the off-target search hot path.
//
// scan_kernel     rows R1-R3 of DESIGN.md: for every read and both strands, every genome window that
//                 the reference's pigeonhole search + verify delegate would accept
//                 (VARSCOT_pipeline/read_mapping/bidir_mapping.cpp:31-148,150-162,285-295).
//                 (the hits are ordered and turned into vsc_hit records by vsc_sort.hip)
// score_kernel    rows R5/R6: calcMitScore (variant_processing/mit_score.h:12-68) and
//                 featureMatrixRecord (variant_processing/feature_matrix.h:25-126) per hit.
//
// Integer / bitwise work on the VALU (v_alignbit, v_xor, v_or, v_bcnt, v_min3, ballot / mbcnt);
// no MFMA: the comparison is not a dense contraction.
#include "vsc_internal.h"
#include "vsc_device.h"

#include <cstring>


namespace vsc {

// ------------------------------------------------------------------------------------------------
// scan kernel
// ------------------------------------------------------------------------------------------------
// A queued site is three words: x = hi plane (bits 0..22) | strand << 23 | edge << 24, l = lo plane,
// pos = global position of the window start.  "edge" marks windows whose successor position is N
// (the only windows the right-edge rule can concern).  Planes are in READ orientation: sites of the
// '-' strand are stored reverse-complemented so that both strands compare against the same read.
struct WaveState {
    uint32_t *qx, *ql, *qpos;  // this wave's LDS site queue
    uint64_t *hkey;            // this wave's LDS hit staging buffer
    uint32_t *hval;
    uint32_t q;       // queue fill (wave-uniform)
    uint32_t hn;      // staged hits (wave-uniform)
    uint32_t lane;
};

// Copies the staged hits of this wave to the global hit arrays (one atomic per flush).
__device__ __forceinline__ void flush_hits(const ScanArgs &a, WaveState &w)
{
    if (w.hn == 0) return;
    unsigned long long base = 0;
    if (w.lane == 0) base = atomicAdd(&a.counters[kCntHits], (unsigned long long)w.hn);
    base = ((unsigned long long)uniform((uint32_t)(base >> 32)) << 32) | uniform((uint32_t)base);
    if (base + w.hn <= a.hit_cap) {
        for (uint32_t i = w.lane; i < w.hn; i += kWave) {
            a.hit_keys[base + i] = w.hkey[i];
            a.hit_vals[base + i] = w.hval[i];
        }
    } else if (w.lane == 0) {
        atomicMax(&a.counters[kCntOverflow], 1ull);  // the host re-runs with a buffer of counters[kCntHits] records
    }
    wave_sync();
    w.hn = 0;
}

// Slow path of the guide loop: stage the hits of one (read, site slot) pair.
// t = mismatch mask in READ orientation, c = popcount(t); lanes with c > max_mm do not take part.
__device__ __forceinline__ void emit_hits(const ScanArgs &a, WaveState &w, uint32_t guide, uint32_t slot, uint32_t t,
                                          uint32_t c)
{
    bool hit = c <= a.max_mm;
    uint32_t pos = 0, strand = 0, mask = 0;
    if (hit) {
        const uint32_t idx = slot * kWave + w.lane;
        const uint32_t x = w.qx[idx];
        pos = w.qpos[idx];
        strand = (x >> kSiteStrandBit) & 1u;
        // mismatch positions in forward-genome window coordinates: on '-' the site was stored
        // reverse-complemented, so read position j is window position 22 - j
        mask = strand ? reverse23(t) : t;
        // Right-edge rule (bidir_mapping.cpp:51-52): a window that ends exactly at its contig's end is
        // never reported through the first-half route; the second-half route needs
        // HD(fullRead[11..23), window[11..23)) <= k.  The separator after a contig is N, so only
        // windows followed by an N position (edge flag) can be affected.
        if ((x >> kSiteEdgeBit) & 1u) {
            if ((uint32_t)__popc(mask >> (VSC_READ_LEN / 2)) > a.k_half &&
                is_contig_end(a.contig_end, a.n_contigs, pos + VSC_READ_LEN))
                hit = false;
        }
    }
    uint64_t b = __ballot(hit);
    if (b == 0) return;
    uint32_t n = (uint32_t)__popcll(b);
    if (hit) {
        uint32_t at = lanes_below(b, w.hn);
        w.hkey[at] = ((uint64_t)guide << 33) | ((uint64_t)strand << 32) | pos;
        w.hval[at] = (c << 23) | mask;
    }
    wave_sync();
    w.hn += n;
    if (w.hn > kHitCap - kWave) flush_hits(a, w);
}

// Drops the first `take` queue entries (what is left, at most 63 entries, moves to the front).
__device__ __forceinline__ void queue_consume(WaveState &w, uint32_t take)
{
    wave_sync();
    const uint32_t left = w.q - take;
    uint32_t vx = 0, vl = 0, vp = 0;
    if (w.lane < left) {
        vx = w.qx[take + w.lane];
        vl = w.ql[take + w.lane];
        vp = w.qpos[take + w.lane];
    }
    wave_sync();
    if (w.lane < left) {
        w.qx[w.lane] = vx;
        w.ql[w.lane] = vl;
        w.qpos[w.lane] = vp;
    }
    wave_sync();
    w.q = left;
}

// Compares the first min(q, kBatch) queued sites of this wave against every read.
__device__ __forceinline__ void process_batch(const ScanArgs &a, WaveState &w)
{
    const uint32_t take = w.q < (uint32_t)kBatch ? w.q : (uint32_t)kBatch;
    uint32_t sh[kSitesPerLane], sl[kSitesPerLane];
#pragma unroll
    for (int j = 0; j < kSitesPerLane; ++j) {
        uint32_t idx = j * kWave + w.lane;
        bool live = idx < take;
        // an empty slot can never match: its upper bits differ from every read's (zero) upper bits,
        // and from the padding reads (all ones) in the lo plane
        sh[j] = live ? (w.qx[idx] & kMask23) : 0xFFFFFFFFu;
        sl[j] = live ? w.ql[idx] : 0u;
    }
    // Reads are fetched through the constant address space, i.e. as scalar loads into SGPRs.  (Tried:
    // wave-uniform vector loads into VGPRs, because tools/valu_rate.hip shows SGPR operands at half
    // rate in isolation - the whole loop got 7 % slower, three-VGPR v_bitop3 pays for it elsewhere.)
    const const_v4u_ptr gp = (const_v4u_ptr)(uintptr_t)a.guides;
    const uint32_t m = a.max_mm;
    // the read table ends with one extra (never matching) group so that the next group can always be
    // fetched while the current one is compared
    v4u na = gp[0], nb = gp[1];
    for (uint32_t g = 0; g < a.n_guides_padded; g += kGuideUnroll) {
        const v4u ga = na, gb = nb;
        na = gp[(g >> 1) + 2];
        nb = gp[(g >> 1) + 3];
        const uint32_t gh[kGuideUnroll] = {ga.x, ga.z, gb.x, gb.z};
        const uint32_t gl[kGuideUnroll] = {ga.y, ga.w, gb.y, gb.w};
        uint32_t best[kGuideUnroll];
#pragma unroll
        for (int u = 0; u < kGuideUnroll; ++u) {
            uint32_t cm = 32;
#pragma unroll
            for (int j = 0; j < kSitesPerLane; ++j) {
                uint32_t t = (sh[j] ^ gh[u]) | (sl[j] ^ gl[u]);
                cm = min(cm, (uint32_t)__popc(t));
            }
            best[u] = cm;
        }
        uint32_t any = min(min(best[0], best[1]), min(best[2], best[3]));
        if (__ballot(any <= m) != 0) {
            // rare: at least one lane has a hit for one of the four reads
#pragma unroll
            for (int u = 0; u < kGuideUnroll; ++u) {
                if (__ballot(best[u] <= m) == 0) continue;
#pragma unroll
                for (int j = 0; j < kSitesPerLane; ++j) {
                    uint32_t t = (sh[j] ^ gh[u]) | (sl[j] ^ gl[u]);
                    uint32_t c = (uint32_t)__popc(t);
                    if (__ballot(c <= m) != 0) emit_hits(a, w, g + u, j, t, c);
                }
            }
        }
    }
    queue_consume(w, take);
}

// Extract mode: appends the first min(q, kBatch) queued sites to the global site arrays.
__device__ __forceinline__ void flush_sites(const ScanArgs &a, WaveState &w)
{
    const uint32_t take = w.q < (uint32_t)kBatch ? w.q : (uint32_t)kBatch;
    unsigned long long base = 0;
    if (w.lane == 0) base = atomicAdd(&a.counters[kCntHits], (unsigned long long)take);
    base = ((unsigned long long)uniform((uint32_t)(base >> 32)) << 32) | uniform((uint32_t)base);
    if (base + take <= a.hit_cap) {
        for (uint32_t i = w.lane; i < take; i += kWave) {
            a.site_x[base + i] = w.qx[i];
            a.site_l[base + i] = w.ql[i];
            a.site_pos[base + i] = w.qpos[i];
        }
    } else if (w.lane == 0) {
        atomicMax(&a.counters[kCntOverflow], 1ull);
    }
    queue_consume(w, take);
}

// kExtract = false: the streaming search (every PAM-valid window against every read).
// kExtract = true : only the guide-independent front end - writes the PAM-valid, N-free windows of
//                   both strands to site_x / site_l / site_pos (input of the seed index, vsc_seed.hip).
template <bool kExtract>
__global__ __launch_bounds__(kWave *kWavesPerGroup) void scan_kernel(const ScanArgs a)
{
    __shared__ uint32_t s_qx[kWavesPerGroup][kQueueCap];
    __shared__ uint32_t s_ql[kWavesPerGroup][kQueueCap];
    __shared__ uint32_t s_qpos[kWavesPerGroup][kQueueCap];
    __shared__ uint64_t s_hkey[kWavesPerGroup][kExtract ? 1 : kHitCap];
    __shared__ uint32_t s_hval[kWavesPerGroup][kExtract ? 1 : kHitCap];

    const uint32_t wave = threadIdx.x / kWave;
    WaveState w;
    w.lane = threadIdx.x % kWave;
    w.qx = s_qx[wave];
    w.ql = s_ql[wave];
    w.qpos = s_qpos[wave];
    w.hkey = s_hkey[wave];
    w.hval = s_hval[wave];
    w.q = 0;
    w.hn = 0;

    const uint32_t n_chunks = (a.n_tiles + kChunkTiles - 1) / kChunkTiles;
    unsigned long long sites = 0;

    for (;;) {
        uint32_t chunk = 0;
        if (w.lane == 0) chunk = (uint32_t)atomicAdd(&a.counters[kCntChunk], 1ull);
        chunk = uniform(chunk);
        if (chunk >= n_chunks) break;
        const uint32_t tile_end = min((chunk + 1) * (uint32_t)kChunkTiles, a.n_tiles);
        for (uint32_t tile = chunk * kChunkTiles; tile < tile_end; ++tile) {
            // ---- one word of each plane per lane (+ its right neighbour for windows that cross) ----
            const size_t wi = (size_t)tile * kTileWords + w.lane;
            const uint32_t H0 = a.hi[wi], H1 = a.hi[wi + 1];
            const uint32_t L0 = a.lo[wi], L1 = a.lo[wi + 1];
            const uint32_t N0 = a.nm[wi], N1 = a.nm[wi + 1];
            // windows (32 starts per lane) that contain an N: OR of the N plane shifted by 0..22
            const uint64_t nraw = ((uint64_t)N1 << 32) | N0;
            uint64_t nn = nraw;
            nn |= nn >> 1;
            nn |= nn >> 2;
            nn |= nn >> 4;
            nn |= nn >> 8;   // bit i covers positions i .. i+15
            nn |= nn >> 7;   // bit i covers positions i .. i+22
            const uint32_t clean = ~(uint32_t)nn;
            const uint32_t edge = (uint32_t)(nraw >> VSC_READ_LEN);  // bit i: position i+23 is N
            // PAM test, 32 window starts at a time (bidir_mapping.cpp:71-76, 240-247)
            const uint32_t H21 = funnel(H1, H0, 21), L21 = funnel(L1, L0, 21);
            const uint32_t H22 = funnel(H1, H0, 22), L22 = funnel(L1, L0, 22);
            const uint32_t Hs1 = funnel(H1, H0, 1), Ls1 = funnel(L1, L0, 1);
            uint32_t vf = 0, vr = 0;
            for (uint32_t i = 0; i < a.n_pam; ++i) {
                const PamMasks p = a.pam[i];
                // '+': window[21] == a and window[22] == b
                vf |= ~(H21 ^ p.ah) & ~(L21 ^ p.al) & ~(H22 ^ p.bh) & ~(L22 ^ p.bl);
                // '-': window[0] == comp(b) and window[1] == comp(a)
                vr |= (H0 ^ p.bh) & (L0 ^ p.bl) & (Hs1 ^ p.ah) & (Ls1 ^ p.al);
            }
            uint32_t mf = vf & clean, mr = vr & clean;
            const uint32_t base_pos = a.first_pos + tile * (uint32_t)kTileBases + w.lane * 32u;

            // ---- move the valid sites into the wave's queue, at most one per lane per step ----
            for (;;) {
                const bool has = (mf | mr) != 0;
                const uint64_t act = __ballot(has);
                if (act == 0) break;
                const bool is_rev = mf == 0;
                const uint32_t msk = is_rev ? mr : mf;
                const uint32_t b = has ? (uint32_t)__builtin_ctz(msk) : 0u;
                const uint32_t rest = msk & (msk - 1u);
                if (is_rev) mr = rest; else mf = rest;
                uint32_t ph = funnel(H1, H0, b) & kMask23;
                uint32_t pl = funnel(L1, L0, b) & kMask23;
                if (is_rev) {
                    ph = revcomp_plane(ph) | (1u << kSiteStrandBit);
                    pl = revcomp_plane(pl);
                }
                ph |= ((edge >> b) & 1u) << kSiteEdgeBit;
                if (has) {
                    const uint32_t at = lanes_below(act, w.q);
                    w.qx[at] = ph;
                    w.ql[at] = pl;
                    w.qpos[at] = base_pos + b;
                }
                const uint32_t added = (uint32_t)__popcll(act);
                w.q += added;
                sites += added;
                wave_sync();
                if (w.q >= (uint32_t)kBatch) {
                    if (kExtract) flush_sites(a, w); else process_batch(a, w);
                }
            }
        }
    }
    if (w.q > 0) {
        if (kExtract) flush_sites(a, w); else process_batch(a, w);
    }
    if (!kExtract) flush_hits(a, w);
    if (w.lane == 0 && sites) atomicAdd(&a.counters[kCntSites], sites);
}

hipError_t launch_scan(const ScanArgs &args, int n_groups, bool extract, hipStream_t stream)
{
    if (extract)
        hipLaunchKernelGGL(scan_kernel<true>, dim3(n_groups), dim3(kWave * kWavesPerGroup), 0, stream, args);
    else
        hipLaunchKernelGGL(scan_kernel<false>, dim3(n_groups), dim3(kWave * kWavesPerGroup), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard results.  Every shard's records are sorted by (guide, strand, contig, pos) and the
// shards partition the positions in ascending order, so the global order is: for every (guide, strand)
// key, the key's segment of shard 0, then of shard 1, ...  No sort is needed - three small kernels
// find the segments, their destinations, and copy them (coalesced 16-byte moves).
// ------------------------------------------------------------------------------------------------
// bound[s * (K + 1) + k] = first record of shard s whose key (guide << 1 | strand) is >= k
__global__ __launch_bounds__(256) void merge_bounds_kernel(const vsc_hit *in, const uint64_t *shard_off, uint32_t n_shards,
                                                           uint32_t K, uint64_t *bound)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_shards * (K + 1)) return;
    const uint32_t s = (uint32_t)(t / (K + 1)), k = (uint32_t)(t % (K + 1));
    uint64_t lo = shard_off[s], hi = shard_off[s + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint32_t key = (in[mid].guide << 1) | VSC_HIT_STRAND(in[mid].info);
        if (key < k) lo = mid + 1; else hi = mid;
    }
    bound[t] = lo;
}

// key_off[k] = number of records (all shards) with key < k; one workgroup
__global__ __launch_bounds__(1024) void merge_scan_kernel(const uint64_t *bound, uint32_t n_shards, uint32_t K, uint64_t *key_off)
{
    __shared__ uint64_t partial[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (K + 1023) / 1024;
    const uint32_t k0 = min(t * per, K), k1 = min(k0 + per, K);
    uint64_t sum = 0;
    for (uint32_t k = k0; k < k1; ++k)
        for (uint32_t s = 0; s < n_shards; ++s) sum += bound[(uint64_t)s * (K + 1) + k + 1] - bound[(uint64_t)s * (K + 1) + k];
    partial[t] = sum;
    block_sync();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t v = t >= d ? partial[t - d] : 0;
        block_sync();
        partial[t] += v;
        block_sync();
    }
    uint64_t run = partial[t] - sum;
    for (uint32_t k = k0; k < k1; ++k) {
        key_off[k] = run;
        for (uint32_t s = 0; s < n_shards; ++s) run += bound[(uint64_t)s * (K + 1) + k + 1] - bound[(uint64_t)s * (K + 1) + k];
    }
    if (t == 1023) key_off[K] = partial[1023];
}

// one workgroup per (key, shard) segment
__global__ __launch_bounds__(256) void merge_copy_kernel(const vsc_hit *in, const uint64_t *bound, const uint64_t *key_off,
                                                         uint32_t n_shards, uint32_t K, vsc_hit *out)
{
    const uint32_t k = blockIdx.x / n_shards, s = blockIdx.x % n_shards;
    uint64_t dst = key_off[k];
    for (uint32_t p = 0; p < s; ++p) dst += bound[(uint64_t)p * (K + 1) + k + 1] - bound[(uint64_t)p * (K + 1) + k];
    const uint64_t b = bound[(uint64_t)s * (K + 1) + k], e = bound[(uint64_t)s * (K + 1) + k + 1];
    const uint4 *src = (const uint4 *)in;
    uint4 *o = (uint4 *)out;
    for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) o[dst + (i - b)] = src[i];
}

hipError_t launch_merge(const vsc_hit *in, const uint64_t *shard_off_dev, uint32_t n_shards, uint32_t K, uint64_t *bound,
                        uint64_t *key_off, vsc_hit *out, hipStream_t stream)
{
    const uint64_t nb = (uint64_t)n_shards * (K + 1);
    hipLaunchKernelGGL(merge_bounds_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, in, shard_off_dev, n_shards,
                       K, bound);
    hipLaunchKernelGGL(merge_scan_kernel, dim3(1), dim3(1024), 0, stream, (const uint64_t *)bound, n_shards, K, key_off);
    hipLaunchKernelGGL(merge_copy_kernel, dim3(K * n_shards), dim3(256), 0, stream, in, (const uint64_t *)bound,
                       (const uint64_t *)key_off, n_shards, K, out);
    return hipGetLastError();
}

// ---- the 8-byte exchange record (multi-GPU) -------------------------------------------------------------------
// What crosses xGMI per hit: mask (bits 0..22) | global position = contig offset + pos (bits 23..54).  Guide and strand
// are not sent: the sender's records are sorted by (guide, strand), so one count per key (guide << 1 | strand) -
// 8 bytes per read instead of 8 bytes per hit - says which records belong to which key; NM is the popcount of the
// mask; the contig follows from the global position.  The receiver's merge kernel rebuilds the 16-byte vsc_hit.
__global__ __launch_bounds__(256) void xpack_kernel(const vsc_hit *in, uint64_t n, const uint32_t *contig_off, uint64_t *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = ((const uint4 *)in)[i];  // {guide, contig, pos, info}
    out[i] = ((uint64_t)(contig_off[r.y] + r.z) << kRecPosShift) | VSC_HIT_MASK(r.w);
}

// one workgroup per (key, shard) segment: seg = k * n_shards + s; the seg_n[seg] records at ADDRESS seg_src[seg] (every shard's
// records may lie in a buffer of their own) become vsc_hit records [seg_dst, ...) of `out` with guide / strand of key
// first_key + k; seg_side (optional): the address of the segment's 16-bit side values (one per record: the votes of the shard's
// classifier), copied to side_out at the same places
__global__ __launch_bounds__(256) void merge_packed_kernel(const uint64_t *seg_src, const uint64_t *seg_dst, const uint32_t *seg_n,
                                                           uint32_t n_shards, uint32_t first_key, const uint32_t *contig_off,
                                                           const uint32_t *contig_end, uint32_t n_contigs, vsc_hit *out, uint32_t *bad,
                                                           const uint64_t *seg_side, uint16_t *side_out)
{
    const uint32_t seg = blockIdx.x;
    const uint32_t n = seg_n[seg];
    if (n == 0) return;
    const uint32_t key = first_key + seg / n_shards;
    const uint64_t *in = (const uint64_t *)seg_src[seg];
    const uint64_t dst = seg_dst[seg];
    uint4 *o = (uint4 *)out;
    if (seg_side) {
        const uint16_t *side = (const uint16_t *)seg_side[seg];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) side_out[dst + i] = side[i];
    }
    // A segment's records ascend in position, and so do the records a thread takes (every blockDim-th): the contig of
    // a record is found by one binary search for the thread's first record and a forward walk from there on (a few
    // contigs per segment: the walk almost never moves).  The records come from a peer or from the caller: one whose
    // window lies in no contig, or that does not ascend, is counted in *bad (the host refuses the result) and the walk is
    // bounded by the contig table whatever the position says.
    uint32_t c = 0, c_start = 0, c_next = 0, prev = 0, n_bad = 0;
    bool placed = false;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t x = in[i];
        const uint32_t gpos = (uint32_t)(x >> kRecPosShift), mask = (uint32_t)x & kMask23;
        // (the neighbour's record: loaded by the neighbouring lane anyway, one more hit on the same line)
        if (i > 0 && (uint32_t)(in[i - 1] >> kRecPosShift) >= gpos) ++n_bad;  // not strictly ascending inside the segment
        if (!placed || gpos < prev) {
            uint32_t lo = 0, hi = n_contigs;  // last contig that starts at or before gpos
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (contig_off[mid] <= gpos) lo = mid; else hi = mid;
            }
            c = lo;
            placed = true;
            c_start = contig_off[c];
            c_next = c + 1 < n_contigs ? contig_off[c + 1] : 0xFFFFFFFFu;
        }
        prev = gpos;
        while (c + 1 < n_contigs && gpos >= c_next) {
            ++c;
            c_start = c_next;
            c_next = c + 1 < n_contigs ? contig_off[c + 1] : 0xFFFFFFFFu;
        }
        n_bad += gpos < c_start || (uint64_t)gpos + VSC_READ_LEN > contig_end[c];
        o[dst + i] = make_uint4(key >> 1, c, gpos - c_start, ((key & 1u) << 31) | ((uint32_t)__popc(mask) << 23) | mask);
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

hipError_t launch_xpack(const vsc_hit *in, uint64_t n, const uint32_t *contig_off, uint64_t *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(xpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, n, contig_off, out);
    return hipGetLastError();
}

hipError_t launch_key_bounds(const vsc_hit *in, const uint64_t *range_dev, uint32_t K, uint64_t *bound, hipStream_t stream)
{
    hipLaunchKernelGGL(merge_bounds_kernel, dim3((unsigned)((K + 1 + 255) / 256)), dim3(256), 0, stream, in, range_dev, 1u, K, bound);
    return hipGetLastError();
}

hipError_t launch_merge_packed(const uint64_t *seg_src, const uint64_t *seg_dst, const uint32_t *seg_n, uint32_t n_segs, uint32_t n_shards,
                               uint32_t first_key, const uint32_t *contig_off, const uint32_t *contig_end, uint32_t n_contigs, vsc_hit *out,
                               uint32_t *bad, const uint64_t *seg_side, uint16_t *side_out, hipStream_t stream)
{
    if (n_segs == 0) return hipSuccess;
    hipLaunchKernelGGL(merge_packed_kernel, dim3(n_segs), dim3(256), 0, stream, seg_src, seg_dst, seg_n, n_shards, first_key, contig_off,
                       contig_end, n_contigs, out, bad, seg_side, side_out);
    return hipGetLastError();
}

// Fingerprint of the resident planes (index files name the genome they belong to): the sum, over all words of all
// three planes, of a 64-bit mix of (word, plane, index) - commutative, so that the order in which workgroups add
// their partial sums does not matter, and sensitive to every bit and to where it stands.
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void plane_hash_kernel(const uint32_t *hi, const uint32_t *lo, const uint32_t *nm, uint64_t n_words,
                                                         unsigned long long *out)
{
    __shared__ unsigned long long partial[256 / kWave];
    uint64_t sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t n = nm[i];
        // the base planes only count where the position is a base: what lies under an N is not part of the genome
        sum += mix64(((uint64_t)(hi[i] & ~(uint32_t)n) << 32 | (lo[i] & ~(uint32_t)n)) + 0x9e3779b97f4a7c15ull * (3 * i + 1));
        sum += mix64(n + 0x9e3779b97f4a7c15ull * (3 * i + 2));
    }
    for (int d = kWave / 2; d > 0; d >>= 1) sum += (uint64_t)__shfl_down((unsigned long long)sum, d, kWave);
    if (threadIdx.x % kWave == 0) partial[threadIdx.x / kWave] = sum;
    block_sync();
    if (threadIdx.x == 0) {
        unsigned long long s = 0;
        for (int w = 0; w < 256 / kWave; ++w) s += partial[w];
        atomicAdd(out, s);
    }
}

hipError_t launch_plane_hash(const uint32_t *hi, const uint32_t *lo, const uint32_t *nm, uint64_t n_words, unsigned long long *out,
                             hipStream_t stream)
{
    if (n_words == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_words + 255) / 256, 4096);
    hipLaunchKernelGGL(plane_hash_kernel, dim3(blocks), dim3(256), 0, stream, hi, lo, nm, n_words, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// per-hit scores
// ------------------------------------------------------------------------------------------------
// variant_processing/mit_score.h:42 - position weights
__constant__ double kMitWeights[20] = {0,     0,     0.014, 0,     0,     0.395, 0.317, 0,     0.389, 0.079,
                                       0.445, 0.508, 0.613, 0.851, 0.732, 0.828, 0.615, 0.804, 0.685, 0.583};

// calcMitScore (variant_processing/mit_score.h:12-68) on the set bits of `mask` (ascending positions,
// forward-genome window coordinates - merge_output_bam.h:549 passes the MD-derived positions as is).
// fp64 with the reference's operation order; contraction off so no multiply-add is fused.
__device__ double mit_score(uint32_t mask, int *ub)
{
#pragma clang fp contract(off)
    *ub = 0;
    const int n = __popc(mask);
    if (n == 0) return 100.0;  // perfectMatch {-1}, :19-22,38-41
    const int last = 31 - __clz(mask);
    const unsigned nm = last < 20 ? (unsigned)n : (unsigned)n - 1u;  // :26-33
    if (nm == 0) return 100.0;                                       // :38-41
    const double s3 = (double)1 / ((double)nm * (double)nm);        // :35, pow(nm, 2) is exact
    double s1 = 1;
    int dist_sum = 0, prev = 0;
    uint32_t rest = mask;
    for (unsigned i = 0; i < nm; ++i) {  // :48-55
        const int p = __ffs(rest) - 1;
        rest &= rest - 1u;
        double wgt = 0;
        if (p < 20) wgt = kMitWeights[p]; else *ub = 1;  // the reference reads past its 20-entry table here
        s1 *= (1 - wgt);
        if (i > 0) dist_sum += p - prev;
        prev = p;
    }
    double s2;
    if (nm < 2) {  // :57-60
        s2 = 1;
    } else {
        const double avg = (double)dist_sum / (double)(nm - 1u);  // :63
        s2 = 1 / (((19 - avg) / 19) * 4 + 1);                      // :64
    }
    return s1 * s2 * s3 * 100;  // :66
}

// featureMatrixRecord (variant_processing/feature_matrix.h:25-126): on / off are 23-base plane pairs
// in read orientation.  f points at 442 zeroed bytes.
__device__ void feature_row(uint32_t on_h, uint32_t on_l, uint32_t off_h, uint32_t off_l, uint8_t *f)
{
    bool prec = false;
    for (int i = 0; i < VSC_READ_LEN - 2; ++i) {  // :53
        const int b = (int)(((off_h >> i) & 1u) << 1 | ((off_l >> i) & 1u));
        if (i < 19) {  // :56-61
            const int b2 = (int)(((off_h >> (i + 1)) & 1u) << 1 | ((off_l >> (i + 1)) & 1u));
            const int pair = b * 4 + b2;
            f[120 + i * 16 + pair] = 1;
            f[424 + pair]++;
        }
        f[36 + i * 4 + b] = 1;  // :64-83
        const int o = (int)(((on_h >> i) & 1u) << 1 | ((on_l >> i) & 1u));
        if (o != b) {  // :86
            f[0]++;
            f[i + 1] = 1;
            if (i > 7 && i < 20) f[441]++;  // :94-98
            if (prec) f[440]++;              // :100-103
            prec = true;
            // :47 transitions AG, CT, GA, TC <=> the two codes differ in the hi bit only
            if ((o ^ b) == 2) f[34]++; else f[35]++;
            // :45-46 AC,AG,AT,CA,CG,CT,GA,GC,GT,TA,TC,TG -> 0..11
            f[22 + o * 3 + (b > o ? b - 1 : b)] = 1;  // :119
        } else {
            prec = false;
        }
    }
}

// hl[w] = (hi[w], lo[w]): scoring reads the four plane words of a hit with two adjacent 8-byte loads
// (one 64-byte sector 7 times out of 8) instead of four loads from two arrays
__global__ __launch_bounds__(256) void interleave_kernel(const uint32_t *hi, const uint32_t *lo, uint64_t n, uint2 *hl)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hl[i] = make_uint2(hi[i], lo[i]);
}

hipError_t launch_interleave(const uint32_t *hi, const uint32_t *lo, uint64_t n, uint2 *hl, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, hi, lo, n, hl);
    return hipGetLastError();
}

// off-target planes of a hit in read orientation: genome[pos, pos+23), reverse-complemented for '-'
// (filter_output_bam.h:399)
__device__ __forceinline__ void site_planes(const ScoreArgs &a, const vsc_hit &h, uint32_t &oh, uint32_t &ol)
{
    const uint32_t rel = a.contig_off[h.contig] + h.pos - a.first_pos;
    const uint64_t wi = rel >> 5;
    const uint32_t sh = rel & 31u;
    const uint2 w0 = a.hl[wi], w1 = a.hl[wi + 1];
    oh = funnel(w1.x, w0.x, sh) & kMask23;
    ol = funnel(w1.y, w0.y, sh) & kMask23;
    if (VSC_HIT_STRAND(h.info)) {
        oh = revcomp_plane(oh);
        ol = revcomp_plane(ol);
    }
}

__global__ __launch_bounds__(256) void score_kernel(const ScoreArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const vsc_hit h = a.hits[i];
    const uint32_t mask = VSC_HIT_MASK(h.info);
    if (a.mit || a.mit_flags) {
        int ub;
        const double s = mit_score(mask, &ub);
        if (a.mit) a.mit[i] = s;
        if (a.mit_flags) a.mit_flags[i] = (uint8_t)ub;
    }
    if (a.features) {
        uint32_t oh, ol;
        site_planes(a, h, oh, ol);
        const uint2 g = a.guides[h.guide];
        uint8_t *f = a.features + i * VSC_N_FEATURES;
        for (int k = 0; k < VSC_N_FEATURES; ++k) f[k] = 0;
        feature_row(g.x, g.y, oh, ol, f);
    }
}

// ------------------------------------------------------------------------------------------------
// packed feature rows: the 442 features of feature_matrix.h:25-126 in 64 bytes per hit
//   w0      bits 0..20 mismatch flags f[1..21] | 21..25 f[0] totalMismatches | 26..30 f[440] adjacentMismatches
//   w1      bits 0..11 mismatch types f[22..33] | 12..16 f[34] transitions | 17..21 f[35] transversions | 22..25 f[441] seed
//   w2..w4  single-letter one-hots f[36..119]  (bit 4 i + base)
//   w5..w14 dinucleotide one-hots f[120..423] (bit 16 i + pair); the 16 dinucleotide counts f[424..439]
//           are the column sums of these flags and are not stored
//   w15     0
// vsc_unpack_features (vsc_pack.cpp) expands a row to the 442 dense values.
// ------------------------------------------------------------------------------------------------
// (feature_row_packed lives in vsc_device.h: the sort's last stage writes the same rows for searches that keep the sites' bases)

// ---- processing order of the packed scoring ------------------------------------------------------------------
// The rows are a 64-byte-per-hit stream, but every hit also gathers its 23-base window from the planes: 16 bytes
// that cost a 64-byte sector each, at random positions of a 0.75 GB array - more traffic than the rows themselves
// (75 B fetched per hit, round 1).  Hits are sorted by (read, strand, position), so the hits of one (read, strand)
// inside a slice of 2^28 positions are one contiguous piece of the result.  Visiting the pieces slice by slice
// keeps the windows of everything in flight inside 64 MB of planes, which the Infinity Cache holds; the rows still
// land at their own index.
//
// bounds[k * (n_slices + 1) + e] = first hit with (read, strand) = k-th pair and position >= e << slice_shift
__global__ __launch_bounds__(256) void score_bounds_kernel(const ScoreArgs a, uint32_t g_lo, uint32_t n_pairs, uint32_t slice_shift,
                                                           uint32_t n_slices, uint64_t *bounds)
{
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint64_t)n_pairs * (n_slices + 1)) return;
    const uint32_t k = (uint32_t)(id / (n_slices + 1)), e = (uint32_t)(id % (n_slices + 1));
    // key = (read << 1 | strand) << 33 | position relative to the shard; e = n_slices: the start of the next pair
    const uint64_t want = e == n_slices ? (uint64_t)(2u * g_lo + k + 1u) << 33 : ((uint64_t)(2u * g_lo + k) << 33) | ((uint64_t)e << slice_shift);
    uint64_t lo = 0, hi = a.n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const vsc_hit h = a.hits[mid];
        const uint64_t key = ((uint64_t)(2u * h.guide + VSC_HIT_STRAND(h.info)) << 33) | (uint64_t)(a.contig_off[h.contig] + h.pos - a.first_pos);
        if (key < want) lo = mid + 1; else hi = mid;
    }
    bounds[id] = lo;
}

// segments in slice-major order and the running sum of their lengths; one workgroup
__global__ __launch_bounds__(1024) void score_segments_kernel(const uint64_t *bounds, uint32_t n_pairs, uint32_t n_slices,
                                                             uint64_t *seg_start, uint64_t *seg_prefix)
{
    __shared__ uint64_t partial[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t n_segs = n_pairs * n_slices;
    const uint32_t per = (n_segs + 1023) / 1024;
    const uint32_t j0 = min(t * per, n_segs), j1 = min(j0 + per, n_segs);
    auto length = [&](uint32_t j) {
        const uint32_t s = j / n_pairs, k = j % n_pairs;
        return bounds[(size_t)k * (n_slices + 1) + s + 1] - bounds[(size_t)k * (n_slices + 1) + s];
    };
    uint64_t sum = 0;
    for (uint32_t j = j0; j < j1; ++j) sum += length(j);
    partial[t] = sum;
    block_sync();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint64_t v = t >= d ? partial[t - d] : 0;
        block_sync();
        partial[t] += v;
        block_sync();
    }
    uint64_t run = partial[t] - sum;
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t s = j / n_pairs, k = j % n_pairs;
        seg_start[j] = bounds[(size_t)k * (n_slices + 1) + s];
        seg_prefix[j] = run;
        run += length(j);
    }
    if (t == 1023) seg_prefix[n_segs] = partial[1023];
}

hipError_t launch_score_schedule(const ScoreArgs &args, uint32_t g_lo, uint32_t g_hi, uint32_t slice_shift, uint32_t n_slices,
                                 uint64_t *bounds, uint64_t *seg_start, uint64_t *seg_prefix, hipStream_t stream)
{
    const uint32_t n_pairs = 2u * (g_hi - g_lo + 1u);
    const uint64_t n_bounds = (uint64_t)n_pairs * (n_slices + 1);
    hipLaunchKernelGGL(score_bounds_kernel, dim3((unsigned)((n_bounds + 255) / 256)), dim3(256), 0, stream, args, g_lo, n_pairs,
                       slice_shift, n_slices, bounds);
    hipLaunchKernelGGL(score_segments_kernel, dim3(1), dim3(1024), 0, stream, (const uint64_t *)bounds, n_pairs, n_slices, seg_start,
                       seg_prefix);
    return hipGetLastError();
}

// One thread per hit computes the row; the rows of a wave (64 x 64 bytes) then go through LDS so that every
// store instruction writes 1 KB of consecutive addresses (a lane storing its own row would write 16 bytes into
// each of 64 different 64-byte sectors per instruction - the kernel is a 64-bytes-per-hit stream).
__global__ __launch_bounds__(256) void score_packed_kernel(const ScoreArgs a, uint4 *packed)
{
    __shared__ uint4 s_rows[256][4];
    __shared__ uint64_t s_index[256];  // the row every thread computed
    __shared__ uint32_t s_count[2];
    const uint32_t t = threadIdx.x;
    const uint64_t v = (uint64_t)blockIdx.x * 256 + t;  // position in processing order
    uint64_t i = v;
    if (a.seg_prefix) {
        // The segment of the workgroup's first position v0 (the last j with seg_prefix[j] <= v0): a 256-ary search by
        // all threads, three rounds of one global load each for 10^5 segments.  (Thread 0 alone, seventeen dependent
        // loads deep while 255 threads waited at the barrier, was a third of the kernel's time: c5 scoring 525 ms.)
        const uint64_t v0 = (uint64_t)blockIdx.x * 256;
        uint32_t lo = 0, hi = a.n_segs;  // seg_prefix[lo] <= v0 < seg_prefix[hi]
        if (t < 2) s_count[t] = 0;
        block_sync();
        for (uint32_t round = 0; hi - lo > 1; ++round) {
            const uint32_t step = (hi - lo + 255u) / 256u;
            const uint32_t at = lo + t * step;
            const bool below = t > 0 && at < hi && a.seg_prefix[at] <= v0;  // true for t = 1 .. k, false from k + 1 on
            const uint64_t b = __ballot(below);
            if ((t & 63u) == 0 && b) atomicAdd(&s_count[round & 1u], (uint32_t)__popcll(b));
            if (t == 0) s_count[(round + 1u) & 1u] = 0;
            block_sync();
            const uint32_t k = s_count[round & 1u];
            lo += k * step;
            hi = min(hi, lo + step);
            block_sync();  // everybody has read the count before the round after next clears it
        }
        if (v < a.n) {
            uint32_t j = lo;
            while (v >= a.seg_prefix[j + 1]) ++j;  // a workgroup's 256 positions span one to three segments
            i = a.seg_start[j] + (v - a.seg_prefix[j]);
        }
    }
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = 0;
    if (v < a.n) {
        const vsc_hit h = a.hits[i];
        uint32_t oh, ol;
        site_planes(a, h, oh, ol);
        const uint2 g = a.guides[h.guide];
        feature_row_packed(g.x, g.y, oh, ol, w);
        if (a.mit) {
            int ub;
            a.mit[i] = mit_score(VSC_HIT_MASK(h.info), &ub);
            if (a.mit_flags) a.mit_flags[i] = (uint8_t)ub;
        }
    }
    s_index[t] = v < a.n ? i : ~0ull;
    // quarter q of row t sits in slot q ^ (t & 3): neighbouring lanes then write to different banks
#pragma unroll
    for (int q = 0; q < 4; ++q) s_rows[t][q ^ (t & 3u)] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    wave_sync();  // a wave only reads back its own 64 rows
    const uint32_t wave_row0 = t & ~63u, lane = t & 63u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t e = k * 64 + lane;  // 16-byte piece e of the wave's 4 KB
        const uint32_t row = wave_row0 + (e >> 2), q = e & 3u;
        const uint64_t at = s_index[row];
        if (at != ~0ull) {  // nontemporal: the rows are not read again here, the planes' lines should stay cached (-2 %)
            const uint4 rv = s_rows[row][q ^ (row & 3u)];
            v4u hv;
            hv.x = rv.x;
            hv.y = rv.y;
            hv.z = rv.z;
            hv.w = rv.w;
            __builtin_nontemporal_store(hv, (v4u *)packed + (at * 4 + q));
        }
    }
}

hipError_t launch_score_packed(const ScoreArgs &args, uint4 *packed, hipStream_t stream)
{
    if (args.n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((args.n + 255) / 256);
    hipLaunchKernelGGL(score_packed_kernel, dim3(blocks), dim3(256), 0, stream, args, packed);
    return hipGetLastError();
}

// The same two scores for explicit (read, site) pairs: planes of both 23-mers in read orientation and
// the mismatch mask the MIT score is taken of (the mergers pass the MD-derived positions).
__global__ __launch_bounds__(256) void score_pairs_kernel(const uint2 *on, const uint2 *off, const uint32_t *masks,
                                                          uint64_t n, double *mit, uint8_t *mit_flags, uint8_t *features)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mit || mit_flags) {
        int ub;
        const double s = mit_score(masks[i] & kMask23, &ub);
        if (mit) mit[i] = s;
        if (mit_flags) mit_flags[i] = (uint8_t)ub;
    }
    if (features) {
        uint8_t *f = features + i * VSC_N_FEATURES;
        for (int k = 0; k < VSC_N_FEATURES; ++k) f[k] = 0;
        feature_row(on[i].x, on[i].y, off[i].x, off[i].y, f);
    }
}

hipError_t launch_score_pairs(const uint2 *on, const uint2 *off, const uint32_t *masks, uint64_t n, double *mit,
                              uint8_t *mit_flags, uint8_t *features, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(score_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, on, off, masks, n, mit,
                       mit_flags, features);
    return hipGetLastError();
}

hipError_t launch_score(const ScoreArgs &args, hipStream_t stream)
{
    if (args.n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((args.n + 255) / 256);
    hipLaunchKernelGGL(score_kernel, dim3(blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// random-forest inference (classification/classificationPipeline.R:27-34, randomForest's classForest):
// x[var] <= split ? left : right from the root until a terminal node, for every tree; votes for class "1".
//
// A workgroup takes 256 feature rows, one thread each.  The forest tests at most 128 distinct columns of the
// 442 (rfClassifier: 79): the workgroup first copies those columns of its rows into LDS - from dense rows, or
// decoded straight from the 64-byte packed rows of vsc_score_hits_packed, so that a result can be classified
// without ever expanding it - then walks the trees tile by tile: as many whole trees as fit 48 KB of LDS are
// staged with coalesced loads, every thread walks them from LDS (a node visit = one 16-byte LDS read + one byte).
// Few rows: gridDim.y splits the trees over several workgroups per row tile, votes meet in an atomic add.
// ------------------------------------------------------------------------------------------------
// The row the test extraction reads (kRfRowWords words): the 16 packed words + the 16 dinucleotide counts (column
// sums of the dinucleotide flags, 5 bits each: words 16, 17 hold six counts, word 18 four) + the activity rank.
__device__ __forceinline__ void rf_row_words(const uint32_t (&w)[16], uint32_t rank, uint32_t (&r)[kRfRowWords])
{
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = w[k];
    uint32_t cnt[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) cnt[p] = 0;
#pragma unroll
    for (int i = 0; i < 19; ++i) {  // the 16 flags of position i are bits 16 i .. 16 i + 15 of words 5..14
        const uint32_t half = (w[5 + (i >> 1)] >> ((i & 1) * 16)) & 0xFFFFu;
#pragma unroll
        for (int p = 0; p < 16; ++p) cnt[p] += (half >> p) & 1u;
    }
    r[16] = r[17] = r[18] = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) r[16 + p / 6] |= cnt[p] << (5 * (p % 6));
    r[19] = rank;
}

// kMode 0: dense rows, 1: packed rows, 2: rows computed here from the hits of a.score (score -> classify fused)
template <int kMode> __global__ __launch_bounds__(kRfRows) void rf_predict_kernel(const RfArgs a)
{
    extern __shared__ uint32_t s_dyn[];
    // s_bits[word][thread]: the row's test results, one bit per test; then the tree tile
    const uint32_t n_words = (a.n_tests + 31u) / 32u;
    const uint32_t t = threadIdx.x;
    // pair form: two planes of 256 threads, eight words of 1 KB each (the walk masks a word's number into the address in place:
    // the words must lie a power of two apart that the node has room for); else words of kRfRows threads
    const bool pairs = a.compact == 2u;
    static_assert(kRfRows == 512 && kRfPairBitsBytes == (kRfRows / 256) * 8192, "pair form: two planes of 256 rows, 8 KB each");
    const uint32_t bits_stride = pairs ? 256u : (uint32_t)kRfRows, bits_base = pairs ? (t >> 8) * 2048u + (t & 255u) : t;
    // test i's bit: word i / 32, bit i % 32 - pair form: word i / 29, bit 3 + i % 29 (the walk takes a daughter's bit with one
    // shift to bit 3, where it is worth 8, half of the exit byte's shift: kRfPairFirstBit)
    const uint32_t first_pos = pairs ? (uint32_t)kRfPairFirstBit : 0u;
    uint32_t *const s_bits = s_dyn;
    uint32_t *const s_tile = s_dyn + (pairs ? (size_t)kRfPairBitsBytes / sizeof(uint32_t) : (size_t)n_words * kRfRows);
    const uint64_t row = (uint64_t)blockIdx.x * kRfRows + t;
    const bool live = row < a.n;
    if (kMode == 0) {
        // dense rows (the command-line tool's path): one byte load per test
        const uint32_t rank = live ? a.act_rank[row] : 0u;
        uint32_t bits = 0, wd = 0, pos = first_pos;
        for (uint32_t i = 0; i < a.n_tests; ++i) {
            const RfTest ts = a.tests[i];
            const uint32_t x = !live ? 0u : (ts.dense_col == VSC_N_FEATURES ? rank : a.dense[row * VSC_N_FEATURES + ts.dense_col]);
            bits |= (x <= ts.thr ? 1u : 0u) << pos;
            if (++pos == 32u) {
                s_bits[wd++ * bits_stride + bits_base] = bits;
                bits = 0;
                pos = first_pos;
            }
        }
        if (pos != first_pos) s_bits[wd * bits_stride + bits_base] = bits;
    } else {
        uint32_t w[16] = {};
        uint32_t rank = 0;
        if (live) {
            if (kMode == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 v = a.packed[row * 4 + q];
                    w[4 * q] = v.x, w[4 * q + 1] = v.y, w[4 * q + 2] = v.z, w[4 * q + 3] = v.w;
                }
                rank = a.act_rank[row];
            } else {
                // the row of hit `row`, as score_packed_kernel computes it - kept in registers, never stored
                const vsc_hit h = a.score.hits[row];
                uint32_t oh, ol;
                site_planes(a.score, h, oh, ol);
                const uint2 g = a.score.guides[h.guide];
                feature_row_packed(g.x, g.y, oh, ol, w);
                rank = a.act_rank[h.guide];
                if (a.score.mit) {
                    int ub;
                    a.score.mit[row] = mit_score(VSC_HIT_MASK(h.info), &ub);
                    if (a.score.mit_flags) a.score.mit_flags[row] = (uint8_t)ub;
                }
            }
        }
        uint32_t r[kRfRowWords];
        rf_row_words(w, rank, r);
        // the tests are sorted by the row word they read: a compile-time loop over the words (registers), a wave-uniform
        // loop over each word's tests (scalar loads), the bits collected in test order
        uint32_t bits = 0, i = 0, wd = 0, pos = first_pos;  // (all four wave-uniform: scalar registers)
#pragma unroll
        for (int k = 0; k < kRfRowWords; ++k) {
            const uint32_t rk = r[k];
            const uint32_t end = a.test_begin[k + 1];
            for (; i < end; ++i) {
                const RfTest ts = a.tests[i];
                const uint32_t x = (rk >> ts.shift) & ((1u << ts.width) - 1u);
                bits |= (x <= ts.thr ? 1u : 0u) << pos;
                if (++pos == 32u) {
                    s_bits[wd++ * bits_stride + bits_base] = bits;
                    bits = 0;
                    pos = first_pos;
                }
            }
        }
        if (pos != first_pos) s_bits[wd * bits_stride + bits_base] = bits;
    }
    // this workgroup's share of the trees
    const uint32_t per_split = (a.n_trees + a.tree_splits - 1) / a.tree_splits;
    const uint32_t tree_begin = blockIdx.y * per_split, tree_end = min(tree_begin + per_split, a.n_trees);
    const uint32_t tile_trees = max(1u, (pairs ? (uint32_t)(kRfPairTileBytes / sizeof(uint2)) : (uint32_t)(kRfTileBytes / sizeof(uint32_t))) / a.n_nodes);
    const uint32_t *const bt = s_bits + bits_base;
    uint32_t ones = 0;
    for (uint32_t t0 = tree_begin; t0 < tree_end; t0 += tile_trees) {
        const uint32_t nt = min(tile_trees, tree_end - t0);
        block_sync();  // the previous tile (first round: the rows' words) is done with
        const uint32_t *src = a.nodes + (size_t)t0 * a.n_nodes;
        if (a.compact == 2u) {  // pair nodes (8 bytes): the same layout as the compact form, [trees of chain 0][sink][trees of chain 1][sink]
            const uint2 *const src2 = (const uint2 *)a.nodes + (size_t)t0 * a.n_nodes;
            uint2 *const tile2 = (uint2 *)s_tile;
            const uint32_t per_nodes = ((nt + kRfChains - 1) / kRfChains) * a.n_nodes, total = nt * a.n_nodes;
            for (uint32_t i = t; i < total; i += kRfRows) {
                uint32_t c = 0;
#pragma unroll
                for (int k = 1; k < kRfChains; ++k) c += i >= (uint32_t)k * per_nodes ? 1u : 0u;
                tile2[i + c] = src2[i];
            }
            if (t < (uint32_t)kRfChains) tile2[min((t + 1u) * per_nodes, total) + t] = make_uint2(0u, 0u);
        } else if (a.compact) {  // [trees of chain 0][sink][trees of chain 1][sink] - see below
            // chain c's trees lie c words further on: a sink word follows every chain
            const uint32_t per_nodes = ((nt + kRfChains - 1) / kRfChains) * a.n_nodes, total = nt * a.n_nodes;
            for (uint32_t i = t; i < total; i += kRfRows) {
                uint32_t c = 0;  // the chain word i belongs to (no division: a handful of compares)
#pragma unroll
                for (int k = 1; k < kRfChains; ++k) c += i >= (uint32_t)k * per_nodes ? 1u : 0u;
                s_tile[i + c] = src[i];
            }
            if (t < (uint32_t)kRfChains) {
                const uint32_t end = min((t + 1u) * per_nodes, total);  // words of the trees of chains 0 .. t
                s_tile[end + t] = 0u;
            }
        } else {
            for (uint32_t i = t; i < nt * a.n_nodes; i += kRfRows) s_tile[i] = src[i];
        }
        block_sync();
        if (a.compact == 2u) {
            // PAIR nodes: the per-lane tree queues of the compact form below, two levels per step.  A node of 8 bytes holds a
            // split node's test, both daughters' tests and the four exits (granddaughters): one 8-byte LDS read - a random
            // ds_read_b64 of a wave costs what a random ds_read_b32 costs, 64 banks of pairs against 32 - and two reads of the
            // row's test words (conflict-free) per TWO levels, instead of two and two.  That leaves
            // the vector unit as the bound (tools/micro/valu_kinds.hip: 2.4 cycles per SIMD for the two-operand shifts / and / add,
            // 2.7 for v_bitop3, 4.2 for every other three-operand form), so the node is laid out for few and cheap instructions:
            //   lower word: exit[2 * root bit + daughter bit], a byte each: vote | pair nodes to skip << 1
            //   upper word: three tests, each as (bit in the row's test word: 5 bits at s, number of the word: 3 bits at s + 10),
            //   s = 0 root, 5 right daughter, 18 left daughter (bit set = x <= thr = left).  A test word's address is the upper
            //   word (shifted by s) masked IN PLACE, OR-ed into the lane's base - one v_bitop3; the root's bit comes out as a mask
            //   (v_bfe_i32, which takes its offset from the low five bits of the same register) that selects the daughter's test
            //   and half of the exit byte's shift; the daughter's bit is shifted to bit 3, where it is the other half (its field
            //   holds the position - 3: test bits lie at positions 3 .. 31 of their words).  (Taking the root's bit by a shift as
            //   well needs the daughters' fields 16 bits apart and a third beside them: no arrangement of three (5 + 3)-bit fields
            //   with the word number a fixed distance above the shift fits 32 bits.)  A terminal daughter: both its exits carry
            //   its vote to the next tree's root.  All zero = the sink (a real node has no zero exit).
            typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
            typedef uint32_t v2u __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(3))) v2u *lds_u64;
            const uint32_t per = (nt + kRfChains - 1) / kRfChains;
            // (bits_at = plane * 8 KB + 4 * row in the plane: the dynamic LDS starts at 0 in this kernel - there is no static LDS -,
            // so bits 10..12, where the word number is OR-ed in, are clear)
            const uint32_t tile_at = (uint32_t)(uintptr_t)(lds_u32)s_tile, bits_at = (uint32_t)(uintptr_t)(lds_u32)bt;
            uint32_t k_word = 0x1C00u, k_16 = 16u;  // (in vector registers: a scalar or literal operand makes v_bitop3 a 4.2-cycle instruction)
            asm volatile("" : "+v"(k_word), "+v"(k_16));
            uint32_t at[kRfChains];
            v2u n[kRfChains];
            uint32_t any = 0;
#pragma unroll
            for (int c = 0; c < kRfChains; ++c) {
                const uint32_t first = min((uint32_t)c * per, nt);
                at[c] = tile_at + (first * a.n_nodes + (uint32_t)c) * 8u;
                n[c] = *(lds_u64)(uintptr_t)at[c];
                any |= n[c].x;
            }
            while (__ballot(any != 0u)) {  // (two steps per chain between the exit tests: a finished chain spins on its sink)
                any = 0;
#pragma unroll
                for (int c2 = 0; c2 < 2 * kRfChains; ++c2) {
                    const int c = c2 % kRfChains;
                    const uint32_t hi = n[c].y;
                    const uint32_t w_root = *(lds_u32)(uintptr_t)__builtin_amdgcn_bitop3_b32(hi, k_word, bits_at, 0xEA);  // (hi & k) | base
                    const uint32_t m_root = (uint32_t)__builtin_amdgcn_sbfe((int)w_root, hi, 1u);                          // all ones: left
                    const uint32_t next = __builtin_amdgcn_bitop3_b32(m_root, hi >> 18, hi >> 5, 0xCA);                    // m ? left : right
                    const uint32_t w_next = *(lds_u32)(uintptr_t)__builtin_amdgcn_bitop3_b32(next, k_word, bits_at, 0xEA);
                    // (a daughter's field holds its bit's position - 3: one shift puts the bit where it is worth 8)
                    const uint32_t d = n[c].x >> __builtin_amdgcn_bitop3_b32(m_root, k_16, (w_next >> (next & 31u)) & 8u, 0xEA);  // exit byte 2 r + d
                    uint32_t vote, skip;  // (opaque: the compiler would extract the vote from the node word with a v_bfe_u32, 4.2 cycles against 2.4)
                    asm("v_and_b32 %0, 1, %1" : "=v"(vote) : "v"(d));
                    asm("v_and_b32 %0, 0xfe, %1" : "=v"(skip) : "v"(d));
                    ones += vote;
                    at[c] += skip << 2;
                    n[c] = *(lds_u64)(uintptr_t)at[c];
                    if (c2 >= kRfChains) any |= n[c].x;
                }
            }
            continue;
        }
        if (a.compact) {
            // Per-lane tree queues: a lane that reaches a terminal node goes straight on to its next tree instead of
            // idling until the deepest tree of the wave is through (the trees are 11 - 21 levels deep, a row's path 9
            // on average).  A node names, per daughter, how many nodes further on the walk continues - to the daughter, or
            // (terminal daughter: never read, its vote travels in the field) to the root of the next tree, which lies right
            // behind this one: one add per step, no tree base to keep.  kRfChains chains per lane - consecutive parts of the tile - overlap
            // their LDS latencies (c5 batch, 256 rows: one chain 8.8 s, two 5.2 s, three 5.9 s; 512 rows and a 26 KB tile: 4.8 s); behind the last tree of each part lies a SINK word (0 = test 0, both daughters "0 nodes
            // on, no vote": it points at itself and votes nothing), so a finished chain spins without side effects
            // and the wave leaves when every lane's two nodes are sinks.  Straight-line code, no per-lane predicates.
            //   tile layout: [trees of chain 0][sink][trees of chain 1][sink] ...
            const uint32_t per = (nt + kRfChains - 1) / kRfChains;  // trees per chain (the last chains may have fewer, or none)
            // (LDS addresses as 32-bit integers: pointer arithmetic on generic pointers costs 64-bit adds and flat loads;
            // the two opaque one-instruction asm statements keep the compiler from re-folding "field, then shift-and-add"
            // into shift + mask + add - three instructions instead of two, twice per step)
            typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
            const uint32_t tile_at = (uint32_t)(uintptr_t)(lds_u32)s_tile, bits_at = (uint32_t)(uintptr_t)(lds_u32)bt;
            uint32_t at[kRfChains];  // LDS address of the chain's current node
            uint32_t n[kRfChains];   // ... and its word
            uint32_t any = 0;
#pragma unroll
            for (int c = 0; c < kRfChains; ++c) {
                const uint32_t first = min((uint32_t)c * per, nt);  // chain c walks trees [first, min(first + per, nt))
                at[c] = tile_at + (first * a.n_nodes + (uint32_t)c) * 4u;  // (c sink words lie in front of it)
                n[c] = *(lds_u32)(uintptr_t)at[c];
                any |= n[c];
            }
            while (__ballot(any != 0u)) {
                any = 0;
#pragma unroll
                for (int c = 0; c < kRfChains; ++c) {
                    // the row's word that holds the node's test (word index = test / 32; words lie kRfRows apart)
                    uint32_t wi;
                    asm("v_bfe_u32 %0, %1, 5, 5" : "=v"(wi) : "v"(n[c]));
                    const uint32_t w = *(lds_u32)(uintptr_t)(bits_at + (wi << 11));
                    // x <= thr (bit set) takes the left daughter, filed at bits 21..31; else the right one at bits 10..20
                    const uint32_t d = __builtin_amdgcn_ubfe(n[c], 10u + 11u * __builtin_amdgcn_ubfe(w, n[c], 1u), 11u);
                    ones += d >> 10;  // the vote bit is set for terminal daughters only
                    uint32_t skip;    // split daughter: that many nodes on; terminal: the next tree's root
                    asm("v_and_b32 %0, 0x3ff, %1" : "=v"(skip) : "v"(d));
                    at[c] += skip << 2;
                    n[c] = *(lds_u32)(uintptr_t)at[c];
                    any |= n[c];
                }
            }
            continue;
        }
        // (forests with more than 512 nodes per tree) Two trees in flight per thread: a walk is a chain of dependent LDS
        // reads, two chains overlap their latencies.  Terminal nodes point at themselves, so every lane simply takes as
        // many steps as the deeper of the two trees is deep (wave-uniform trip count: no per-lane exit test).
        for (uint32_t k = 0; k < nt; k += 2) {
            const uint32_t k1 = min(k + 1, nt - 1);
            const uint32_t *const tree0 = s_tile + (size_t)k * a.n_nodes, *const tree1 = s_tile + (size_t)k1 * a.n_nodes;
            const uint32_t steps = max((uint32_t)a.depth[t0 + k], (uint32_t)a.depth[t0 + k1]);
            uint32_t n0 = tree0[0], n1 = tree1[0];
            for (uint32_t step = 0; step < steps; ++step) {
                const uint32_t b0 = (bt[((n0 & 1023u) >> 5) * kRfRows] >> (n0 & 31u)) & 1u;
                const uint32_t b1 = (bt[((n1 & 1023u) >> 5) * kRfRows] >> (n1 & 31u)) & 1u;
                n0 = tree0[(n0 >> (b0 ? 10u : 20u)) & 1023u];
                n1 = tree1[(n1 >> (b1 ? 10u : 20u)) & 1023u];
            }
            ones += n0 >> 31;
            if (k + 1 < nt) ones += n1 >> 31;
        }
    }
    if (!live) return;
    if (kMode == 2)
        a.votes16[row] = (uint16_t)ones;
    else if (a.tree_splits > 1)
        atomicAdd(&a.votes[row], ones);
    else
        a.votes[row] = ones;
}

hipError_t launch_rf_predict(const RfArgs &args, hipStream_t stream)
{
    if (args.n == 0) return hipSuccess;
    if (args.n_tests > (uint32_t)kRfMaxTests || args.n_tests == 0 || args.n_nodes > (uint32_t)kRfMaxNodes || args.n_nodes == 0 ||
        (args.compact == 2u ? (size_t)args.n_nodes * sizeof(uint2) > (size_t)kRfPairTileBytes || args.n_tests > (uint32_t)kRfPairMaxTests || args.n_nodes > 127u
                            : (size_t)args.n_nodes * sizeof(uint32_t) > (size_t)kRfTileBytes))
        return hipErrorInvalidValue;
    const uint64_t tiles = (args.n + kRfRows - 1) / kRfRows;
    if (tiles >= (1ull << 31)) return hipErrorInvalidValue;
    const size_t n_words = (args.n_tests + 31) / 32;
    // test bits + the tree tile (+ a sink node per chain)
    const size_t lds = args.compact == 2u ? (size_t)kRfPairBitsBytes + kRfPairTileBytes + 8 * kRfChains
                                          : n_words * kRfRows * sizeof(uint32_t) + (size_t)kRfTileBytes + 4 * kRfChains;
    const int mode = args.dense ? 0 : (args.packed ? 1 : 2);
    if (mode == 2 && args.tree_splits != 1) return hipErrorInvalidValue;
    auto go = [&](auto kernel) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)tiles, args.tree_splits), dim3(kRfRows), lds, stream, args);
        return hipGetLastError();
    };
    if (mode == 0) return go(rf_predict_kernel<0>);
    if (mode == 1) return go(rf_predict_kernel<1>);
    return go(rf_predict_kernel<2>);
}

}  // namespace vsc
