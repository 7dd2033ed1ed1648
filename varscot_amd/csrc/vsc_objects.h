// vsc_objects.h - the objects behind the opaque handles of include/varscot_hip.h, shared by the host-side
// translation units of the library (vsc_api.cpp, vsc_multi.cpp).  Not installed.
#pragma once

#include <string>
#include <vector>

#include "varscot_hip_debug.h"
#include "vsc_internal.h"

namespace vsc {

struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    // hipMalloc / hipFree of multi-GB buffers cost hundreds of milliseconds: grow with 1/8 headroom so
    // that result sizes that wobble from search to search do not reallocate every time
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes < ((size_t)32 << 30) ? bytes / 8 : bytes / 32);  // (3 % on buffers of tens of GB: batch sizes wobble by 1 %)
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess && want != bytes) {
            (void)hipGetLastError();
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

inline vsc_debug_params default_debug_params()
{
    vsc_debug_params d{};
    d.sort_xcd = -1;
    d.sort_optimistic = -1;
    d.score_slices = -1;
    d.seed_shared = -1;
    d.seed_group_out = -1;
    d.seed_tight = -1;
    d.rf_form = -1;
    return d;
}

// process-wide: host-side lap times on stderr (vsc_debug_set_host_timing)
bool host_timing_on();

}  // namespace vsc

struct vsc_ctx {
    int device = 0;
    int n_cus = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::string err;
    vsc_timing timing{};
    vsc_debug_params dbg = vsc::default_debug_params();  // test / experiment hooks (include/varscot_hip_debug.h)
    // keys_a / keys_b: the two record buffers the bin sort alternates between (keys_a + vals_a: the scan's (key, value) pairs)
    vsc::DeviceBuf counters, guides, keys_a, keys_b, vals_a, score_mit, score_flags, score_feat;
    vsc::DeviceBuf vals_b;  // side words beside keys_b (searches that keep the sites' bases; vals_a then holds those beside keys_a)
    // the read planes of the scoring calls (their own buffer: `guides` is the search passes'), kept while the read set stays the same
    vsc::DeviceBuf score_guides;
    uint64_t score_guides_hash = 0;
    uint32_t score_guides_n = ~0u;
    vsc::DeviceBuf sort_segs, sort_tabs, sort_over, score_sched;
    // host staging of the sort's segment tables: uploaded with hipMemcpyAsync, so they must outlive the call that enqueues them
    std::vector<vsc::SortSeg> host_segs;
    std::vector<uint32_t> host_tile0;  // bin sort: segment table + tile starts, per-bin tables, oversize list + counter
    vsc::DeviceBuf seed_off, seed_poff, seed_lrest;  // per-search read lists: bucket counts, padded list starts, entries
    // the forest of the last classification call, as the kernels read it (prepare_forest in vsc_api.cpp)
    struct Forest {
        vsc::DeviceBuf nodes, ranks;   // nodes + tree depths + test table; activity ranks of the reads of a fused call
        size_t depth_at = 0, tests_at = 0, begin_at = 0;
        uint32_t n_tests = 0, n_trees = 0, n_nodes = 0;
        uint32_t form = 0;              // node form (vsc_internal.h): 0 plain, 1 compact, 2 pairs
        uint32_t node_stride = 0;       // nodes from one tree's root to the next (n_nodes; pair form: the largest tree's pair nodes)
        std::vector<double> thresholds;  // distinct activity splits, ascending
        uint64_t fingerprint = 0;
        uint64_t ranks_key = 0;  // activities + forest the resident ranks were made from
        uint32_t ranks_n = ~0u;
    } forest;
    // record buffers of freed results, kept for the next search: hipMalloc / hipFree of tens of GB
    // cost hundreds of milliseconds each
    std::vector<vsc::DeviceBuf> spare_records;
};

struct vsc_genome {
    vsc_ctx *ctx = nullptr;
    uint32_t *d_hi = nullptr, *d_lo = nullptr, *d_nm = nullptr;
    uint32_t *d_contig_off = nullptr, *d_contig_end = nullptr;
    uint2 *d_hl = nullptr;  // interleaved planes, built on first scoring call
    uint64_t first_word = 0, own_words = 0, dev_words = 0;
    uint32_t n_tiles = 0, n_contigs = 0;
    uint64_t device_bytes = 0;
    uint64_t sites = 0;  // PAM-valid windows seen by the last scan (sizes the next hit buffer)
    // hits per read the last search with mismatch budget m produced (scan: all reads; seed: the fullest output
    // region) - real genomes are not the uniform model the first buffer size comes from
    double seen_rate[VSC_MAX_MISMATCHES + 1] = {};
    // the sort's first level may skip its histogram pass (slot partition) until a search with this budget has
    // produced a bin that outgrew its slot (repeats)
    bool sort_slots_ok[VSC_MAX_MISMATCHES + 1] = {true, true, true, true, true, true, true, true, true};
    // seed index (vsc_seed.hip): the PAM-valid sites filed once per segment, sorted by bucket
    bool has_index = false;
    uint8_t index_has_extra_pam = 0;
    char index_extra_pam[2] = {0, 0};
    uint64_t index_sites = 0;  // S
    uint4 *d_ix_chunk_tab = nullptr;
    uint32_t *d_ix_vert = nullptr;  // bit-sliced blocks of 32 sites
    uint2 *d_ix_sites = nullptr;    // 8-byte site records {rest planes, position}
    uint32_t *d_ix_edge = nullptr;  // 1 bit per site: window followed by N
    uint32_t ix_chunks = 0;
    uint64_t ix_class_sites[4] = {};   // sites per PAM class in one table, chunks per class over all three (the search's cost model)
    uint64_t ix_class_chunks[4] = {};
    uint64_t ix_vert_bytes = 0, ix_edge_words = 0;  // sizes of d_ix_vert / d_ix_edge (the index file stores them)
    uint64_t index_bytes = 0;
    double index_ms = 0;
};

struct vsc_hits {
    vsc_ctx *ctx = nullptr;
    vsc_hit *d_records = nullptr;
    vsc::DeviceBuf storage;  // owns d_records
    uint64_t n = 0;
    std::vector<vsc_hit> host;
    bool host_valid = false;
};


namespace vsc {
// A genome object that only carries the contig table (no planes, nothing to search): what vsc_hits_merge_packed
// needs on a device whose shard of a tiny genome is empty (vsc_multi.cpp).  Freed with vsc_genome_free.
int genome_table_only(vsc_ctx *ctx, const vsc_contig *contigs, uint32_t n_contigs, vsc_genome **out);
// vsc_hits_merge_packed over per-shard record buffers on ctx's device, with optional 16-bit side values per record that
// arrive in side_out in merged order (vsc_api.cpp)
int merge_packed_shards(vsc_ctx *ctx, const vsc_genome *genome, const void *const *shard_records, const void *const *shard_side,
                        const uint32_t *key_counts, uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out,
                        DeviceBuf *side_out);
}  // namespace vsc
