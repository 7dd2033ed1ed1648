// vsc_internal.h - types shared by the C-ABI host code (vsc_api.cpp) and the HIP kernels
// (vsc_kernels.hip).  Not installed; the public contract is include/varscot_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "varscot_hip.h"

namespace vsc {

// ---- scan geometry ---------------------------------------------------------------------------
constexpr int kWave = 64;                       // gfx950 wavefront
constexpr int kWavesPerGroup = 4;               // 256-thread workgroups; the 4 waves never synchronise
constexpr int kSitesPerLane = 8;                // candidate sites a lane keeps in VGPRs in the guide loop
constexpr int kBatch = kWave * kSitesPerLane;   // 512 sites per wave per guide-loop pass
constexpr int kQueueCap = kBatch + kWave;       // per-wave LDS site queue (a step adds at most 64)
constexpr int kHitCap = 128;                    // per-wave LDS hit staging buffer (flushed above 64)
constexpr int kTileWords = kWave;               // one 32-base word of each plane per lane = 2048 positions
constexpr int kTileBases = kTileWords * 32;
constexpr int kChunkTiles = 8;                  // tiles a wave takes per grab of the work counter
constexpr int kGuideUnroll = 4;                 // guides per inner iteration (guide table is padded to it)
constexpr uint32_t kMask23 = 0x7FFFFFu;
constexpr int kSiteStrandBit = 23;              // site word x: bit 23 = '-' strand
constexpr int kSiteEdgeBit = 24;                // site word x: bit 24 = the position after the window is N
constexpr uint32_t kPadWords = 4;               // words the device planes are padded with past the shard

// ---- seed-partitioned search (vsc_seed.hip) -------------------------------------------------------
constexpr int kSegments = 3;                    // read positions [0,7) [7,14) [14,21); the PAM is in no segment
constexpr int kSegBases = 7;
constexpr int kBucketsPerSeg = 1 << (2 * kSegBases);  // 16384 seven-mers
constexpr int kBuckets = kSegments * kBucketsPerSeg;
constexpr int kSeedHitCap = 256;                // smallest block of records a wave reserves per atomic
constexpr int kSlicedSites = 32;                // sites per lane of the bit-sliced comparison (one bit each)
constexpr int kSlicedChunk = kWave * kSlicedSites;  // 2048 sites per wave and chunk
constexpr int kRestBases = VSC_READ_LEN - kSegBases;  // 16 read positions outside the seed segment (what a site record keeps)
constexpr int kCmpBases = 2 * kSegBases;        // ... of which the comparison counts 14: the other two segments.  Read positions
                                                // 21 and 22 are the PAM, and a chunk holds sites of ONE PAM (its class)
constexpr int kVertWords = 2 * kCmpBases;       // words of a bit-sliced block of 32 sites: hi and lo plane bit of the 14 positions
constexpr int kSeedClasses = 3;                 // PAMs an index can hold: GG, GA (+ -P); class of a site = which one its window ends in
// read lists: one per bucket for segments 0 and 1 (shared by the classes - an entry holds a budget per class), one per
// (class, bucket) for segment 2, whose neighbourhood depends on what the class leaves of the mismatch limit (seed_enum_kernel)
constexpr int kLists = (kSegments - 1 + kSeedClasses) * kBucketsPerSeg;
constexpr int kListBudgetShift = 16;            // list entry y: read index | per class c: mismatches left for the 14 compared positions
constexpr uint32_t kListNoBudget = 15;          //   << (16 + 4 c), 15 = the read has no business with that class (list padding: all ones)
constexpr int kTokLaneShift = 26;               // sliced hit token, high word: read of the pass (14 bits) | chunk slot << 14 | lane << 26
constexpr int kTokSlotShift = 14;
constexpr uint32_t kTokReadMask = (1u << kTokSlotShift) - 1u;
constexpr int kSlicedResolve = 3;                // sliced kernel: resolve when this many passes of 64 tokens wait (the
                                                // gathers of the later passes overlap the earlier ones)
constexpr int kSlicedTokCap = 512;  // per-wave LDS ring of 8-byte hit tokens (a power of two: the slot is an AND): a group of four reads adds <= 256
// chunk table word z: bucket (16 bits) | rank in the chunk of its first '-' site (0 .. 2048) << 16 | "holds a window
// that is followed by N" << 28 | class of its sites << 29
constexpr uint32_t kChunkBucketMask = 0xFFFFu;
constexpr int kChunkMinusShift = 16;
constexpr int kChunkEdgeBit = 28;
constexpr int kChunkClassShift = 29;
constexpr uint32_t kIndexLayout = 2;            // seed index layout (index files of another layout are refused)
constexpr int kSlicedWavesPerSimd = 6;          // resident waves of the sliced kernel per SIMD (registers and LDS allow five)
constexpr int kSlicedGrab = 8;                  // chunks of 2048 sites per grab of the work counter (sliced kernel)

// counters[] slots of one scan launch
enum { kCntHits = 0, kCntChunk = 1, kCntSites = 2, kCntOverflow = 3, kCntVisited = 4, kCntPad = 5, kCntLost = 6, kCntGroups = 7, kCntSlots = 8 };
// The sliced kernel hands out chunks through kCursors work cursors, one per slice of the chunk range, 128
// bytes apart behind the counters: one device-wide cursor sustains only ~90 atomics/us
// (MI355X_MICROARCH.md), which for 270 000 grabs is 3 ms - as long as a whole 1 000-read search.
// The sliced kernel writes its hits into up to kParts regions by read range (region = read index >>
// kRegionBits, 64 reads): inside a region a hit is one packed 8-byte record (below), and the regions
// are ordered independently by the bin sort of vsc_sort.hip; their concatenation is the result.
// Per region: counters[kCntPart + 4 p + {0, 1, 2}] = records reserved, sentinels among them, records lost.
constexpr int kRegionBits = 6;
constexpr int kRegionReads = 1 << kRegionBits;
constexpr int kParts = 256;
constexpr int kMaxPassReads = kParts * kRegionReads;  // reads one search pass can take (vsc_search splits larger sets)
constexpr int kCntPart = 8;
constexpr int kCursors = 32;
constexpr int kCursorStride = 16;  // in 8-byte words
constexpr int kCursorBase = kCntPart + 4 * kParts + 8;  // first cursor, in 8-byte words from the start of the counter buffer
constexpr int kCounterWords = kCursorBase + kCursors * kCursorStride;

// ---- packed hit record (what the search kernels hand to the sort) --------------------------------------
//   bits  0..22  mismatch mask in forward-genome window coordinates (NM is its popcount)
//   bits 23..54  position of the window start RELATIVE TO THE SHARD's first position (a shard of the upper half of the
//                genome would otherwise leave the top key bits constant: bins by the top position bits fill unevenly, slots
//                overflow, levels are wasted), shifted left by pos_pad = 32 - (bits the shard's positions need): the zero
//                bits of a small genome sit at the bottom of the field, where the sort's last stage ignores them, instead
//                of in the middle of the key, where partition levels would be spent on them
//   bit  55      strand (1 = '-')
//   bits 56..61  read index inside its region (read & 63)
//   bits 62, 63  0; the all-ones word is the sentinel that pads reserved-but-unused record slots
// Ascending order of the records of one region = the result order (read, strand, position).
constexpr int kRecPosShift = 23;
constexpr int kRecStrandShift = 55;
constexpr int kRecReadShift = 56;
constexpr int kRecKeyBits = 33 + kRegionBits;  // sort key = rec >> kRecPosShift
constexpr uint64_t kRecSentinel = ~0ull;

// ---- bin sort (vsc_sort.hip) -------------------------------------------------------------------------
constexpr int kSortThreads = 512;                      // partition: 512 threads x 16 records = a tile of 8 192 in 64 KB of LDS, two workgroups per CU
                                                      // (one tile of 16 384 per CU: 6.3-6.5 ms per c3 partition against 5.9-6.0)
constexpr int kSortItems = 16;
constexpr int kSortTile = kSortThreads * kSortItems;  // records per partition tile
constexpr int kFinThreads = 512;                      // finalize kernel: 512 threads x 16 records, two workgroups per CU
constexpr int kFinItems = 16;
constexpr int kSortCap = kFinThreads * kFinItems - 128;  // the largest bin it orders in LDS (records + sub-bin table = half
                                                      // of the CU's 160 KiB, so that one workgroup computes while the other
                                                      // waits for memory)
constexpr int kSortMaxBinBits = 11;                   // <= 2048 bins per partition level
constexpr int kSortSubBits = 13;                      // <= 8 192 sub-bins inside the finalize kernel
constexpr int kSortSubBitsRows = 12;                  // ... 4 096 when it also writes the feature rows: the other half of the table's LDS
                                                      // stages the rows on their way out
constexpr int kHistTiles = 8;                         // tiles a block of the histogram kernel walks through

// Required plane bits of the two PAM letters, expanded to all-ones / all-zero words.
struct PamMasks {
    uint32_t ah, al;  // first letter  (window position 21 on '+')
    uint32_t bh, bl;  // second letter (window position 22 on '+')
};

struct ScanArgs {
    const uint32_t *hi, *lo, *nm;  // device planes; element 0 = the shard's first word; padded by kPadWords
    uint32_t first_pos;            // global position of bit 0 of element 0
    uint32_t n_tiles;              // tiles whose window starts are searched
    const uint4 *guides;           // two reads per uint4: (hi0, lo0, hi1, lo1); count padded to kGuideUnroll
    uint32_t n_guides_padded;
    uint32_t max_mm;               // -M
    uint32_t k_half;               // floor(max_mm / 2): the per-half budget of the reference's pigeonhole search
    uint32_t n_pam;
    PamMasks pam[3];               // GG, GA (+ -P)
    const uint32_t *contig_end;    // ascending global end positions (offset + length) of all contigs
    uint32_t n_contigs;
    uint64_t *hit_keys;            // out: guide << 33 | strand << 32 | global position
    uint32_t *hit_vals;            // out: NM << 23 | mismatch mask (window coordinates)
    unsigned long long hit_cap;
    unsigned long long *counters;  // kCntSlots values, zeroed before the launch
    // extract mode only: PAM-valid sites of both strands (x = hi plane | strand | edge, l = lo plane);
    // the append cursor is counters[kCntHits], the capacity hit_cap
    uint32_t *site_x, *site_l, *site_pos;
};

// One segment of the bin sort: a span of packed records that is ordered independently of all others (level 1:
// one output region of the search kernel; deeper levels: one bin that was too large for the finalize kernel).
struct SortSeg {
    uint64_t in_off;     // first record of the segment in the level's source buffer
    uint64_t out_off;    // first record of the segment's span in the level's destination buffer
    uint64_t final_off;  // index of the segment's first vsc_hit in the result
    uint32_t n_in;       // records in the source span (level 1: sentinels included)
    uint32_t guide_base; // read index of the region's first read
};

struct SortArgs {
    const SortSeg *segs;
    const uint32_t *seg_tile0;     // [n_segs + 1] first tile of every segment (tile = kSortTile records)
    uint32_t n_segs, n_tiles;
    const uint64_t *in;            // packed records
    const uint64_t *pair_keys;     // level 0 only (streaming scan): guide << 33 | strand << 32 | position ...
    const uint32_t *pair_vals;     //                                ... and NM << 23 | mask; `in` is then unused
    uint64_t *out;
    const uint32_t *side_in;       // null, or one 32-bit side word per record of `in` (the site's lo plane) ...
    uint32_t *side_out;            // ... moved to the same places of this array as the records in `out`
    uint32_t *hist;                // [n_segs << bin_bits] records per bin
    uint32_t *cursor;              // [n_segs << bin_bits] next free record of every bin (relative to the segment)
    uint32_t *bin_start;           // [n_segs << bin_bits] first record of every bin (relative to the segment)
    uint32_t bin_bits, bin_shift;  // bin = (record >> bin_shift) & ((1 << bin_bits) - 1)
    uint32_t pos_pad;              // level 0: left shift of the position field
    uint32_t pos_base;             // level 0: first global position of the shard (records hold positions relative to it)
    uint32_t xcd_tiles;            // partition: tiles per XCD (0: workgroup b takes tile b)
    // slot mode (no histogram pass): bin i of the level owns records [i * slot_cap, (i + 1) * slot_cap) of `out`; the
    // partition reserves room with `cursor` (zeroed) alone and raises *overflow when a bin does not fit its slot -
    // the caller then repeats the level with the histogram.  0: exact mode (bin_start from the histogram)
    uint32_t slot_cap;
    uint32_t *overflow;
};

struct FinArgs {
    const SortSeg *segs;
    uint32_t n_segs;
    const uint64_t *src;           // the buffer the last partition level wrote (or the search kernel, if none ran)
    const uint32_t *hist;          // bins of the last partition level; null: every segment is one bin (its source span)
    const uint32_t *bin_start;
    uint32_t bin_bits;
    uint32_t sub_shift, sub_bits;  // LDS counting sort on (record >> sub_shift) & ((1 << sub_bits) - 1) ...
    uint32_t low_bits;             // ... then ranking on the key bits below (0: none left)
    uint32_t pos_pad;              // left shift of the records' position field
    uint32_t pos_base;             // first global position of the shard: the records' positions count from it
    SortSeg *over;                 // bins with more than `cap` records are listed here for another level
    uint32_t over_cap;
    uint32_t cap;                  // <= kSortCap (smaller in tests only)
    uint32_t *n_over;
    uint32_t *cursor;              // zeroed: the resident workgroups count the bins they take here
    const uint32_t *contig_off;    // ascending global start positions of all contigs
    uint32_t n_contigs;
    vsc_hit *out;
    uint32_t slot_cap;             // slot mode: bin i's records are src[i * slot_cap ...) (hist = the partition's cursors)
    const uint32_t *overflow;      // slot mode: non-zero = the partition gave up, nothing to do
    // a search that keeps the sites' bases (SeedArgs.hit_side): the side words beside `src`, the reads' planes, and where the
    // 64-byte packed feature row of result record i goes: rows + (i - rows_first) * 64 bytes
    const uint32_t *side_src;
    const uint2 *guides;           // (hi plane, lo plane) of read guide_first + j
    uint32_t guide_first;
    uint4 *rows;                   // null: plain records (low 23 bits = mismatch mask)
    uint64_t rows_first;
};

struct ScoreArgs {
    const vsc_hit *hits;  // device records, already offset to the first row to score
    uint64_t n;
    const uint2 *hl;          // interleaved (hi, lo) plane words of the shard that holds the hits
    uint32_t first_pos;
    uint64_t n_plane_words;
    const uint32_t *contig_off;
    const uint2 *guides;  // (hi plane, lo plane) per read
    double *mit;          // may be null
    uint8_t *mit_flags;   // may be null
    uint8_t *features;    // may be null; n * 442 bytes
    // optional processing order (score_packed_kernel): the rows are visited genome slice by genome slice so that
    // the window gathers of everything in flight fall into one slice of the planes (see launch_score_schedule);
    // virtual position v lies in segment j (seg_prefix[j] <= v < seg_prefix[j + 1]) and is row seg_start[j] + v - seg_prefix[j]
    const uint64_t *seg_prefix;  // [n_segs + 1]; null: rows in index order
    const uint64_t *seg_start;   // [n_segs]
    uint32_t n_segs;
};

// How one search cuts the pigeonhole (seed_enum_kernel).  A site of class c (its PAM) leaves a read left_c = max_mm -
// (mismatches of the read's last two letters with that PAM) for read positions 0..20.  Segment 0 is searched within k0
// substitutions, segment 1 within k1, segment 2 within left_c - k0 - k1 - 2 (not at all if negative): a window that fails all
// three has at least (k0 + 1) + (k1 + 1) + (left_c - k0 - k1 - 1) = left_c + 1 mismatches.  Every (k0, k1) in 0..2 that keeps
// the third threshold <= 2 is a valid cut; the host picks one by cost (vsc_api.cpp).  tight = 0: the round-3 cut -
// floor(max_mm / 3) in all three segments.
struct SeedPlan {
    uint32_t max_mm, k0, k1, tight;
    uint32_t n_nbr;      // neighbours enumerated per (read, segment): 1 / 22 / 211 = the largest threshold in use
    uint32_t n_pam;
    uint32_t pam_codes;  // class c: (first letter << 2 | second letter) << 4 c
};

struct SeedArgs {
    const uint4 *chunk_tab;        // [n_chunks] {first site, site count, bucket | first '-' rank << 16 | edge << 28 | class << 29, first vertical block}
    const uint32_t *vert;          // bit-sliced copies of the sites: kVertWords words per block of 32 sites (see seed_transpose_kernel)
    const uint2 *list_rest;        // per list entry {rest(hi) | rest(lo) << 16, read | budget per class << 16 ..}
    const uint2 *sites;            // {rest(hi) | rest(lo) << 16, position} per site
    const uint32_t *edge_bits;     // 1 bit per site: its window is followed by N
    const uint2 *guides;           // (hi plane, lo plane) per read
    uint32_t n_chunks;
    const uint32_t *poff;          // [kLists + 1] first entry of every read list (multiples of kGuideUnroll)
    uint32_t max_mm, k_half;
    uint32_t k_seg0, k_seg1;       // substitutions searched in segments 0 and 1 (SeedPlan.k0, .k1): what the duplicate rule tests
    const uint32_t *contig_end;
    uint32_t n_contigs;
    uint64_t *hit_recs;            // out: packed records (layout above), region p = [p * part_cap, (p + 1) * part_cap)
    uint32_t *hit_side;            // null, or (per-hit feature rows wanted): one word beside every record = the site's lo plane in read
                                   // orientation; its hi plane then sits in the record's low 23 bits in place of the mismatch mask
    uint32_t group_out;            // 1: the four waves of a workgroup share their open blocks (chunk-sharing kernel only)
    uint32_t reserve;              // records a wave reserves per atomic on its region's cursor: a power of two, 64 .. 1024
    uint32_t reserve_log2;
    uint32_t pos_pad;              // left shift of the position field of a record
    uint32_t pos_base;             // first global position of the shard: a record holds position - pos_base
    uint32_t n_parts;              // regions in use (region of a hit = read index >> kRegionBits)
    unsigned long long part_cap;   // records per region
    unsigned long long *counters;  // kCntPart + 4 p + {0,1,2}, kCntSites (= pairs compared), kCntVisited, kCntOverflow, cursors
};

// ---- random-forest inference ---------------------------------------------------------------------------------
// Every split of the forest is a TEST `x <= thr` on a small non-negative integer: the predictors of the feature
// matrix are flags and counts (`x <= split` is `x <= floor(split)`), and the on-target activity (a double, constant
// per read) enters as its rank among the forest's distinct activity thresholds (rank = thresholds strictly below it),
// which turns `activity <= T_j` into `rank <= j` exactly.  The forest has few DISTINCT tests (rfClassifier: 217 for
// 105 235 split nodes), so a row is reduced to one bit per test first (7 words), and a node is 4 bytes:
//   bits 0..9 test | 10..19 left daughter | 20..29 right daughter (0-based; a terminal node points at itself) |
//   30 terminal | 31 votes class "1"
// A node visit is one 4-byte LDS read of the node + one of the row's test word.
// Trees of at most 512 nodes (rfClassifier: 275) use the COMPACT form, in which a node says of each daughter how many
// nodes further on the walk continues - at the daughter (which lies behind its parent), or, if the daughter is terminal,
// at the root of the NEXT tree (trees lie back to back) with the terminal node's vote in the field: terminal nodes are never
// visited, a lane moves on to its next tree the moment it reaches one, and a step is one add:
//   bits 0..9 test | 10..19 right: nodes to skip, 20 its vote | 21..30 left: nodes to skip, 31 vote
// (the all-zero word is the SINK behind the last tree of a chain: it skips nothing and votes nothing)
constexpr int kRfRows = 512;             // feature rows per workgroup (one thread each)
constexpr int kRfMaxTests = 1024;        // distinct (predictor, threshold) pairs a forest may use
constexpr int kRfMaxNodes = 1024;        // nodes of one tree (rfClassifier: 275)
constexpr int kRfTileBytes = 26624;      // whole trees staged in LDS per step (with 14 KB of test bits: 4 workgroups = 32 waves per CU)
constexpr int kRfPairFirstBit = 3;       // pair form: test bits lie at positions 3 .. 31 of their words (29 per word) ...
constexpr int kRfPairMaxTests = 8 * (32 - kRfPairFirstBit);  // ... of which a row has eight: 232 tests
constexpr int kRfPairBitsBytes = 16384;  // pair form: the test bits as two planes of 256 rows x 8 words
constexpr int kRfPairTileBytes = 24512;  // ... and its tree tile (16 + 24 KB: four workgroups per CU)
constexpr int kRfChains = 2;             // trees a thread walks at the same time (compact form)
constexpr int kRfRowWords = 20;          // a row as the test extraction sees it: 16 packed words, 3 words of dinucleotide
                                         // counts (5 bits each, 6 / 6 / 4), 1 word activity rank
// a test: field `width` bits at `shift` of row word `word` <= thr; dense rows read column `dense_col` instead
struct RfTest {
    uint8_t word, shift, width, thr;
    uint16_t dense_col;  // 0..441, or 442 = the activity rank
    uint16_t pad;
};

struct RfArgs {
    const uint32_t *nodes;      // [n_trees * n_nodes], tree-major
    const uint8_t *depth;       // [n_trees] steps from the root to the deepest terminal node
    uint32_t n_trees, n_nodes;
    uint32_t compact;           // 2: pair nodes (8 bytes, two levels each), 1: compact nodes (<= 512 per tree), both walked through
                                // per-lane tree queues; 0: the form above
    const RfTest *tests;        // [n_tests], sorted by the row word they read
    const uint32_t *test_begin; // [kRfRowWords + 1] first test of every row word
    uint32_t n_tests;
    // the rows: dense (n x 442 bytes), packed (n x 64 bytes), or - both null - computed in the kernel from `score`'s
    // hits (the fused score -> classify path: the feature rows never exist in memory)
    const uint8_t *dense;
    const uint4 *packed;
    const uint8_t *act_rank;    // activity rank per row (dense / packed) or per read (fused)
    uint64_t n;
    uint32_t *votes;            // [n] trees voting class "1" (zeroed before the launch when tree_splits > 1) ...
    uint16_t *votes16;          // ... or 16-bit (fused path; tree_splits == 1)
    uint32_t tree_splits;       // gridDim.y: every workgroup row walks n_trees / tree_splits trees
    ScoreArgs score;            // fused path: hits, planes, reads (+ optional MIT output)
};

// Launch wrappers implemented in vsc_kernels.hip.  They only enqueue work on `stream`.
hipError_t launch_scan(const ScanArgs &args, int n_groups, bool extract, hipStream_t stream);
// vsc_sort.hip
hipError_t launch_bin_hist(const SortArgs &args, hipStream_t stream);
hipError_t launch_bin_scan(const SortArgs &args, hipStream_t stream);
hipError_t launch_bin_partition(const SortArgs &args, hipStream_t stream);
hipError_t launch_bin_finalize(const FinArgs &args, int max_groups, hipStream_t stream);
hipError_t launch_rf_predict(const RfArgs &args, hipStream_t stream);
hipError_t launch_interleave(const uint32_t *hi, const uint32_t *lo, uint64_t n, uint2 *hl, hipStream_t stream);
hipError_t launch_score(const ScoreArgs &args, hipStream_t stream);
hipError_t launch_score_packed(const ScoreArgs &args, uint4 *packed, hipStream_t stream);
// Fills seg_start / seg_prefix for the hits of args (sorted by guide, strand, position; guides g_lo .. g_hi occur):
// segments = (slice of 2^slice_shift positions, guide, strand), slice-major.  bounds: scratch of
// (g_hi - g_lo + 1) * 2 * (n_slices + 1) 64-bit words.
hipError_t launch_score_schedule(const ScoreArgs &args, uint32_t g_lo, uint32_t g_hi, uint32_t slice_shift, uint32_t n_slices,
                                 uint64_t *bounds, uint64_t *seg_start, uint64_t *seg_prefix, hipStream_t stream);
hipError_t launch_score_pairs(const uint2 *on, const uint2 *off, const uint32_t *masks, uint64_t n, double *mit,
                              uint8_t *mit_flags, uint8_t *features, hipStream_t stream);
// 8-byte exchange records (vsc_hits_pack_exchange / vsc_hits_merge_packed)
hipError_t launch_xpack(const vsc_hit *in, uint64_t n, const uint32_t *contig_off, uint64_t *out, hipStream_t stream);
// bound[k] = first record of in[range_dev[0] .. range_dev[1]) whose key (guide << 1 | strand) is >= k, k = 0 .. K
hipError_t launch_key_bounds(const vsc_hit *in, const uint64_t *range_dev, uint32_t K, uint64_t *bound, hipStream_t stream);
// seg_src / seg_side: ADDRESSES of every (key, shard) segment's records / 16-bit side values (seg_side, side_out: null = none)
hipError_t launch_merge_packed(const uint64_t *seg_src, const uint64_t *seg_dst, const uint32_t *seg_n, uint32_t n_segs, uint32_t n_shards,
                               uint32_t first_key, const uint32_t *contig_off, const uint32_t *contig_end, uint32_t n_contigs, vsc_hit *out,
                               uint32_t *bad /* zeroed */, const uint64_t *seg_side, uint16_t *side_out, hipStream_t stream);
// *out += fingerprint of words [0, n_words) of the three planes (zero *out first)
hipError_t launch_plane_hash(const uint32_t *hi, const uint32_t *lo, const uint32_t *nm, uint64_t n_words, unsigned long long *out,
                             hipStream_t stream);
// vsc_seed.hip
hipError_t launch_seed_keys(const uint4 *rec, uint64_t n, int seg, uint32_t pam_codes, uint32_t n_pam, uint64_t *sort_records, hipStream_t stream);
hipError_t launch_seed_lists(const uint2 *guides, uint32_t n_guides, const SeedPlan &plan, uint32_t *count, uint32_t *poff, uint2 *list_rest,
                             hipStream_t stream);
hipError_t launch_seed_pack16(const uint32_t *x, const uint32_t *l, const uint32_t *pos, uint64_t n, uint4 *rec, hipStream_t stream);
hipError_t launch_seed_gather16(const uint4 *rec, const uint64_t *sorted, uint64_t n, uint4 *out, hipStream_t stream);
hipError_t launch_seed_compact(const uint4 *sites16, uint64_t n_per_table, uint64_t n, uint2 *sites8, uint32_t *edge_bits,
                               hipStream_t stream);
hipError_t launch_seed_chunk_flags(uint4 *chunk_tab, uint32_t n_chunks, const uint32_t *edge_bits, hipStream_t stream);
hipError_t launch_seed_transpose(const uint4 *sites, const uint4 *chunk_tab, uint32_t n_chunks, uint32_t *vert,
                                 hipStream_t stream);
hipError_t launch_seed_sliced(const SeedArgs &args, int n_groups, bool shared, hipStream_t stream);
hipError_t launch_merge(const vsc_hit *in, const uint64_t *shard_off_dev, uint32_t n_shards, uint32_t K, uint64_t *bound,
                        uint64_t *key_off, vsc_hit *out, hipStream_t stream);

}  // namespace vsc
