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
constexpr uint32_t kPadWords = 4;               // words the device planes are padded with past the shard

// counters[] slots of one scan launch
enum { kCntHits = 0, kCntChunk = 1, kCntSites = 2, kCntOverflow = 3, kCntSlots = 4 };

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
};

struct FinalizeArgs {
    const uint64_t *keys;
    const uint32_t *vals;
    uint64_t n;
    const uint32_t *contig_off;  // ascending global start positions of all contigs
    uint32_t n_contigs;
    vsc_hit *out;
};

struct ScoreArgs {
    const vsc_hit *hits;  // device records, already offset to the first row to score
    uint64_t n;
    const uint32_t *hi, *lo;  // planes of the shard that holds the hits
    uint32_t first_pos;
    uint64_t n_plane_words;
    const uint32_t *contig_off;
    const uint2 *guides;  // (hi plane, lo plane) per read
    double *mit;          // may be null
    uint8_t *mit_flags;   // may be null
    uint8_t *features;    // may be null; n * 442 bytes
};

// Launch wrappers implemented in vsc_kernels.hip.  They only enqueue work on `stream`.
hipError_t launch_scan(const ScanArgs &args, int n_groups, hipStream_t stream);
hipError_t sort_temp_bytes(uint64_t n, unsigned end_bit, size_t *bytes);
hipError_t launch_sort(void *temp, size_t temp_bytes, const uint64_t *keys_in, uint64_t *keys_out,
                       const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, unsigned end_bit, hipStream_t stream);
hipError_t launch_finalize(const FinalizeArgs &args, hipStream_t stream);
hipError_t launch_score(const ScoreArgs &args, hipStream_t stream);
hipError_t merge_temp_bytes(uint64_t n, unsigned end_bit, size_t *bytes);
hipError_t launch_merge(void *temp, size_t temp_bytes, const vsc_hit *in, uint64_t n, unsigned end_bit, uint32_t *keys_a,
                        uint32_t *keys_b, uint32_t *idx_a, uint32_t *idx_b, vsc_hit *out, hipStream_t stream);

}  // namespace vsc
