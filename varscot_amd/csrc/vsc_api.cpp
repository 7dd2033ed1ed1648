// vsc_api.cpp - the C ABI of include/varscot_hip.h: context, resident genome, search orchestration
// (scan -> radix sort -> record assembly) and per-hit scoring.  Host C++ only; all device work is in
// vsc_kernels.hip.  No CPU implementation of the search exists in this library: without a HIP
// device every compute entry point fails with VSC_ERR_NODEVICE / VSC_ERR_DEVICE.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vsc_internal.h"
#include "vsc_objects.h"

using namespace vsc;

namespace {

// vsc_debug_set_host_timing(1): host-side wall times of the phases of vsc_search on stderr
struct HostTimer {
    bool on = vsc::host_timing_on();
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vsc timing] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

}  // namespace

namespace vsc {
static std::atomic<int> g_host_timing{0};
bool host_timing_on() { return g_host_timing.load(std::memory_order_relaxed) != 0; }
}  // namespace vsc

namespace {

int fail(vsc_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess)
            std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else
            std::snprintf(buf, sizeof buf, "%s", what);
        ctx->err = buf;
    }
    return code;
}

#define VSC_HIP(ctx, call)                                                 \
    do {                                                                   \
        hipError_t e_ = (call);                                            \
        if (e_ != hipSuccess) return fail((ctx), e_ == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, #call, e_); \
    } while (0)

// Scoring reads the genome at random hit positions: give it the interleaved copy of the planes.
hipError_t ensure_hl(vsc_ctx *ctx, const vsc_genome *genome)
{
    vsc_genome *g = const_cast<vsc_genome *>(genome);
    if (g->d_hl) return hipSuccess;
    hipError_t e = hipMalloc((void **)&g->d_hl, g->dev_words * sizeof(uint2));
    if (e != hipSuccess) return e;
    g->device_bytes += g->dev_words * sizeof(uint2);
    return launch_interleave(g->d_hi, g->d_lo, g->dev_words, g->d_hl, ctx->stream);
}

// Record storage for a result: the smallest spare buffer that fits, else a new allocation.
hipError_t take_records(vsc_ctx *ctx, vsc_hits *hits, uint64_t n)
{
    const size_t bytes = n * sizeof(vsc_hit);
    int best = -1;
    for (size_t i = 0; i < ctx->spare_records.size(); ++i)
        if (ctx->spare_records[i].cap >= bytes && (best < 0 || ctx->spare_records[i].cap < ctx->spare_records[best].cap))
            best = (int)i;
    if (best >= 0) {
        hits->storage = ctx->spare_records[best];
        ctx->spare_records.erase(ctx->spare_records.begin() + best);
    } else {
        // nothing fits: drop the spares first so that their memory can be reused
        for (auto &b : ctx->spare_records) b.release();
        ctx->spare_records.clear();
        hipError_t e = hits->storage.ensure(bytes);
        if (e != hipSuccess) return e;
    }
    hits->d_records = (vsc_hit *)hits->storage.p;
    return hipSuccess;
}

// P[Binomial(21, 3/4) <= m]: chance that a PAM-valid random window is within m mismatches of a read
double hit_probability(unsigned m)
{
    double p = 0, c = 1;
    for (unsigned j = 0; j <= m && j <= 21; ++j) {
        if (j > 0) c = c * (21 - (j - 1)) / j;
        p += c * std::pow(0.75, (double)j) * std::pow(0.25, (double)(21 - j));
    }
    return p;
}

PamMasks pam_masks(int a, int b)
{
    PamMasks m;
    m.ah = (a & 2) ? 0xFFFFFFFFu : 0u;
    m.al = (a & 1) ? 0xFFFFFFFFu : 0u;
    m.bh = (b & 2) ? 0xFFFFFFFFu : 0u;
    m.bl = (b & 1) ? 0xFFFFFFFFu : 0u;
    return m;
}

int base_code(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

void guide_planes(uint64_t g, uint32_t *h, uint32_t *l)
{
    uint32_t hh = 0, ll = 0;
    for (int i = 0; i < VSC_READ_LEN; ++i) {
        const unsigned c = (unsigned)(g >> (2 * i)) & 3u;
        hh |= (c >> 1) << i;
        ll |= (c & 1u) << i;
    }
    *h = hh;
    *l = ll;
}

}  // namespace

// No C++ exception crosses the C boundary (include/varscot_hip.h): every entry point that allocates on the host runs
// inside this guard.
template <class F> int guarded(vsc_ctx *ctx, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try {
            if (ctx) ctx->err = "out of host memory";
        } catch (...) {
        }
        return VSC_ERR_NOMEM;
    } catch (...) {
        try {
            if (ctx) ctx->err = "unexpected C++ exception";
        } catch (...) {
        }
        return VSC_ERR_DEVICE;
    }
}


extern "C" {

int vsc_abi_version(void) { return VSC_ABI_VERSION; }

int vsc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vsc_ctx_create(int device_id, vsc_ctx **out) { return vsc_ctx_create_masked(device_id, nullptr, 0, out); }

int vsc_ctx_create_masked(int device_id, const uint32_t *cu_mask, uint32_t n_mask_words, vsc_ctx **out)
{
    if (!out) return VSC_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VSC_ERR_NODEVICE;
    if (device_id < 0 || device_id >= n) return VSC_ERR_INVALID;
    vsc_ctx *ctx = new (std::nothrow) vsc_ctx();
    if (!ctx) return VSC_ERR_NOMEM;
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess ||
        (cu_mask && n_mask_words ? hipExtStreamCreateWithCUMask(&ctx->stream, n_mask_words, cu_mask) : hipStreamCreate(&ctx->stream)) != hipSuccess) {
        delete ctx;
        return VSC_ERR_DEVICE;
    }
    ctx->own_stream = true;
    ctx->n_cus = prop.multiProcessorCount;
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            vsc_ctx_destroy(ctx);
            return VSC_ERR_DEVICE;
        }
    *out = ctx;
    return VSC_OK;
}

int vsc_ctx_release_scratch(vsc_ctx *ctx)
{
    if (!ctx) return VSC_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (DeviceBuf *b : {&ctx->counters, &ctx->guides, &ctx->score_guides, &ctx->keys_a, &ctx->keys_b, &ctx->vals_a, &ctx->vals_b,
                         &ctx->score_mit, &ctx->score_flags, &ctx->score_feat, &ctx->score_sched, &ctx->sort_segs, &ctx->sort_tabs,
                         &ctx->sort_over, &ctx->seed_off,
                         &ctx->seed_poff, &ctx->seed_lrest})
        b->release();
    for (auto &b : ctx->spare_records) b.release();
    ctx->spare_records.clear();
    ctx->forest.nodes.release();
    ctx->forest.ranks.release();
    ctx->forest.fingerprint = 0;
    return VSC_OK;
}

int vsc_ctx_destroy(vsc_ctx *ctx)
{
    if (!ctx) return VSC_OK;
    (void)vsc_ctx_release_scratch(ctx);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VSC_OK;
}

int vsc_ctx_set_stream(vsc_ctx *ctx, void *hip_stream)
{
    if (!ctx) return VSC_ERR_INVALID;
    if (ctx->own_stream && ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return VSC_OK;
}

const char *vsc_last_error(const vsc_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int vsc_ctx_set_debug_params(vsc_ctx *ctx, const vsc_debug_params *params)
{
    if (!ctx) return VSC_ERR_INVALID;
    ctx->dbg = params ? *params : default_debug_params();
    return VSC_OK;
}

int vsc_ctx_get_debug_params(const vsc_ctx *ctx, vsc_debug_params *out)
{
    if (!ctx || !out) return VSC_ERR_INVALID;
    *out = ctx->dbg;
    return VSC_OK;
}

void vsc_debug_set_host_timing(int on) { vsc::g_host_timing.store(on ? 1 : 0, std::memory_order_relaxed); }

int vsc_ctx_timing(const vsc_ctx *ctx, vsc_timing *out)
{
    if (!ctx || !out) return VSC_ERR_INVALID;
    *out = ctx->timing;
    return VSC_OK;
}

// ------------------------------------------------------------------------------------------------
int vsc_genome_load(vsc_ctx *ctx, const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t first_word,
                    uint64_t n_words, uint64_t own_words, const vsc_contig *contigs, uint32_t n_contigs,
                    vsc_genome **out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    ctx->err.clear();
    if (!hi || !lo || !nmask || !contigs || n_contigs == 0 || n_words == 0 || own_words == 0 || own_words > n_words)
        return fail(ctx, VSC_ERR_INVALID, "vsc_genome_load: null or empty argument");
    if (own_words < n_words && own_words % kTileWords != 0)
        return fail(ctx, VSC_ERR_INVALID, "vsc_genome_load: a shard followed by halo words must own a multiple of 64 words");
    const uint64_t n_tiles = (own_words + kTileWords - 1) / kTileWords;
    const uint64_t dev_words = std::max<uint64_t>(n_tiles * kTileWords, n_words) + kPadWords;
    if ((first_word + dev_words) * 32 >= (1ull << 32) - 4096)
        return fail(ctx, VSC_ERR_RANGE, "vsc_genome_load: genome exceeds the 32-bit position space (4 Gbases)");
    std::vector<uint32_t> off(n_contigs), end(n_contigs);
    for (uint32_t c = 0; c < n_contigs; ++c) {
        const uint64_t e = contigs[c].offset + contigs[c].length;
        if (e >= (1ull << 32) - 4096) return fail(ctx, VSC_ERR_RANGE, "vsc_genome_load: contig table exceeds 4 Gbases");
        if (c > 0 && contigs[c].offset < contigs[c - 1].offset + contigs[c - 1].length + 1)
            return fail(ctx, VSC_ERR_INVALID, "vsc_genome_load: contigs must be ascending and separated by >= 1 N position");
        off[c] = (uint32_t)contigs[c].offset;
        end[c] = (uint32_t)e;
    }
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    vsc_genome *g = new (std::nothrow) vsc_genome();
    if (!g) return fail(ctx, VSC_ERR_NOMEM, "vsc_genome_load: out of host memory");
    g->ctx = ctx;
    g->first_word = first_word;
    g->own_words = own_words;
    g->dev_words = dev_words;
    g->n_tiles = (uint32_t)n_tiles;
    g->n_contigs = n_contigs;
    const size_t pb = dev_words * sizeof(uint32_t), cb = (size_t)n_contigs * sizeof(uint32_t);
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    step(hipMalloc((void **)&g->d_hi, pb));
    step(hipMalloc((void **)&g->d_lo, pb));
    step(hipMalloc((void **)&g->d_nm, pb));
    step(hipMalloc((void **)&g->d_contig_off, cb));
    step(hipMalloc((void **)&g->d_contig_end, cb));
    if (e == hipSuccess) {
        step(hipMemsetAsync(g->d_hi, 0, pb, ctx->stream));
        step(hipMemsetAsync(g->d_lo, 0, pb, ctx->stream));
        step(hipMemsetAsync(g->d_nm, 0xFF, pb, ctx->stream));  // everything past the shard is N
        step(hipMemcpyAsync(g->d_hi, hi, n_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        step(hipMemcpyAsync(g->d_lo, lo, n_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        step(hipMemcpyAsync(g->d_nm, nmask, n_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        step(hipMemcpyAsync(g->d_contig_off, off.data(), cb, hipMemcpyHostToDevice, ctx->stream));
        step(hipMemcpyAsync(g->d_contig_end, end.data(), cb, hipMemcpyHostToDevice, ctx->stream));
        step(hipStreamSynchronize(ctx->stream));
    }
    if (e != hipSuccess) {
        vsc_genome_free(g);
        return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, "vsc_genome_load", e);
    }
    g->device_bytes = 3 * pb + 2 * cb;
    *out = g;
    return VSC_OK;
    });
}

}  // extern "C"

int vsc::genome_table_only(vsc_ctx *ctx, const vsc_contig *contigs, uint32_t n_contigs, vsc_genome **out)
{
    if (!ctx || !out || !contigs || n_contigs == 0) return VSC_ERR_INVALID;
    *out = nullptr;
    std::vector<uint32_t> off(n_contigs), end(n_contigs);
    for (uint32_t c = 0; c < n_contigs; ++c) {
        off[c] = (uint32_t)contigs[c].offset;
        end[c] = (uint32_t)(contigs[c].offset + contigs[c].length);
    }
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    vsc_genome *g = new (std::nothrow) vsc_genome();
    if (!g) return fail(ctx, VSC_ERR_NOMEM, "out of host memory");
    g->ctx = ctx;
    g->n_contigs = n_contigs;
    const size_t cb = (size_t)n_contigs * sizeof(uint32_t);
    hipError_t e = hipMalloc((void **)&g->d_contig_off, cb);
    if (e == hipSuccess) e = hipMalloc((void **)&g->d_contig_end, cb);
    if (e == hipSuccess) e = hipMemcpy(g->d_contig_off, off.data(), cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_contig_end, end.data(), cb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        vsc_genome_free(g);
        return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, "contig table upload", e);
    }
    g->device_bytes = 2 * cb;
    *out = g;
    return VSC_OK;
}

extern "C" {

int vsc_genome_free(vsc_genome *g)
{
    if (!g) return VSC_OK;
    if (g->ctx) (void)hipSetDevice(g->ctx->device);
    for (void *p : {(void *)g->d_hi, (void *)g->d_lo, (void *)g->d_nm, (void *)g->d_contig_off, (void *)g->d_contig_end, (void *)g->d_hl,
                    (void *)g->d_ix_chunk_tab, (void *)g->d_ix_vert, (void *)g->d_ix_sites, (void *)g->d_ix_edge})
        if (p) (void)hipFree(p);
    delete g;
    return VSC_OK;
}

uint64_t vsc_genome_device_bytes(const vsc_genome *g) { return g ? g->device_bytes + g->index_bytes : 0; }

// ------------------------------------------------------------------------------------------------
namespace {

#define VSC_TRY(call)                        \
    do {                                     \
        hipError_t e_ = (call);              \
        if (e_ != hipSuccess) return e_;     \
    } while (0)

void fill_pam(ScanArgs &a, const vsc_search_params *params)
{
    a.n_pam = 2;
    a.pam[0] = pam_masks(2, 2);  // GG  (bidir_mapping.cpp:240)
    a.pam[1] = pam_masks(2, 0);  // GA
    if (params && params->has_extra_pam) {  // :242-247
        const int p0 = base_code(params->extra_pam[0]), p1 = base_code(params->extra_pam[1]);
        // a PAM containing a non-ACGT letter can only match windows that contain N, which never
        // pass the verification (:81-82) - it adds nothing
        if (p0 < 4 && p1 < 4) a.pam[a.n_pam++] = pam_masks(p0, p1);
    }
}

// the PAM set as the seed index numbers it: class c = PAM c, (first letter << 2 | second letter) << 4 c
uint32_t pam_code_set(const ScanArgs &a)
{
    uint32_t codes = 0;
    for (uint32_t c = 0; c < a.n_pam; ++c)
        codes |= ((((a.pam[c].ah & 2u) | (a.pam[c].al & 1u)) << 2) | (a.pam[c].bh & 2u) | (a.pam[c].bl & 1u)) << (4 * c);
    return codes;
}

void fill_genome(ScanArgs &a, const vsc_ctx *ctx, const vsc_genome *genome)
{
    a.hi = genome->d_hi;
    a.lo = genome->d_lo;
    a.nm = genome->d_nm;
    a.first_pos = (uint32_t)(genome->first_word * 32);
    a.n_tiles = genome->n_tiles;
    a.contig_end = genome->d_contig_end;
    a.n_contigs = genome->n_contigs;
    a.counters = (unsigned long long *)ctx->counters.p;
}

int scan_groups(const vsc_ctx *ctx, uint32_t n_tiles)
{
    const uint32_t n_waves_max = (uint32_t)ctx->n_cus * 4 * kWavesPerGroup;
    const uint32_t n_chunks = (n_tiles + kChunkTiles - 1) / kChunkTiles;
    const uint32_t n_waves = std::max<uint32_t>(1, std::min<uint32_t>(n_waves_max, n_chunks));
    return (int)((n_waves + kWavesPerGroup - 1) / kWavesPerGroup);
}

bool index_matches(const vsc_genome *g, const vsc_search_params *p)
{
    if (!g->has_index) return false;
    const bool want = p && p->has_extra_pam && base_code(p->extra_pam[0]) < 4 && base_code(p->extra_pam[1]) < 4;
    if (want != (g->index_has_extra_pam != 0)) return false;
    return !want || (base_code(p->extra_pam[0]) == base_code(g->index_extra_pam[0]) &&
                     base_code(p->extra_pam[1]) == base_code(g->index_extra_pam[1]));
}

void free_index(vsc_genome *g)
{
    for (void **p : {(void **)&g->d_ix_chunk_tab, (void **)&g->d_ix_vert, (void **)&g->d_ix_sites, (void **)&g->d_ix_edge}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    g->has_index = false;
    g->index_bytes = 0;
    g->ix_chunks = 0;
}

void class_counts(vsc_genome *g, const std::vector<uint32_t> &ctab);

// Builds the seed index of a resident genome: extract the PAM-valid sites of both strands, then file
// them once per segment in bucket order.  All temporaries are released before returning.
hipError_t build_index(vsc_ctx *ctx, vsc_genome *g, const vsc_search_params *params, std::string *why)
{
    free_index(g);
    hipStream_t st = ctx->stream;
    ScanArgs a{};
    fill_pam(a, params);
    fill_genome(a, ctx, g);
    const int n_groups = scan_groups(ctx, a.n_tiles);
    VSC_TRY(ctx->counters.ensure(kCounterWords * sizeof(unsigned long long)));
    a.counters = (unsigned long long *)ctx->counters.p;
    unsigned long long cnt[kCntSlots];
    VSC_TRY(hipEventRecord(ctx->ev[5], st));
    // pass 1: count (capacity 0: nothing is written, the cursor still counts)
    VSC_TRY(hipMemsetAsync(ctx->counters.p, 0, sizeof cnt, st));
    VSC_TRY(launch_scan(a, n_groups, true, st));
    VSC_TRY(hipMemcpyAsync(cnt, ctx->counters.p, sizeof cnt, hipMemcpyDeviceToHost, st));
    VSC_TRY(hipStreamSynchronize(st));
    const uint64_t S = cnt[kCntHits];
    if (3 * S >= (1ull << 32) - (1u << 20)) {
        *why = "too many PAM-valid sites for 32-bit table indices";
        return hipErrorInvalidValue;
    }
    // Temporaries: the extracted sites (sx, sl, sp), sort scratch, the bucket-ordered sites as 16-byte records
    // (full planes: what the bit-slicing pass reads) and the (bucket, strand) starts; all released before returning.
    DeviceBuf sx, sl, sp, full, rec16;
    auto release = [&]() {
        for (DeviceBuf *b : {&sx, &sl, &sp, &full, &rec16}) b->release();
    };
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    HostTimer ht;
    constexpr uint32_t kKeys = 8 * kBuckets;  // (bucket, class, strand) triples: inside a bucket the sites lie class by class (four
                                              // slots, kSeedClasses in use), inside a class '+' sites precede '-' sites
    const uint32_t pam_codes = pam_code_set(a);
    // Allocations follow the phases, so that the peak stays at 84 bytes per site:
    //   extract   sx, sl, sp (12 B/site) -> rec16 (16)                          then sx, sl, sp go
    //   order     rec16 (16) + two buffers of sort records (16: the context's pooled keys_a / keys_b, which the searches use
    //             afterwards anyway) -> full (48)                               then rec16 goes
    //   compact   full (48) -> the resident 8-byte records (24) + edge bits
    //   slice     full (48) + records (24) -> bit-sliced blocks (12)            then full goes
    const size_t n4 = std::max<uint64_t>(S, 1) * sizeof(uint32_t);
    for (DeviceBuf *b : {&sx, &sl, &sp}) step(b->ensure(n4));
    step(rec16.ensure(std::max<uint64_t>(S, 1) * sizeof(uint4)));
    const size_t edge_words = (size_t)((3 * S + 31) / 32 + 1);
    ht.lap("index: count pass + buffers");
    if (e == hipSuccess && S > 0) {
        // pass 2: emit
        a.site_x = (uint32_t *)sx.p;
        a.site_l = (uint32_t *)sl.p;
        a.site_pos = (uint32_t *)sp.p;
        a.hit_cap = S;
        step(hipMemsetAsync(ctx->counters.p, 0, sizeof cnt, st));
        step(launch_scan(a, n_groups, true, st));
        step(launch_seed_pack16((const uint32_t *)sx.p, (const uint32_t *)sl.p, (const uint32_t *)sp.p, S, (uint4 *)rec16.p, st));
        step(hipStreamSynchronize(st));
    }
    for (DeviceBuf *b : {&sx, &sl, &sp}) b->release();
    ht.lap("index: emit pass");
    // ---- the three tables: the sites ordered by (bucket of the table's segment, strand) -----------------------------------
    // A counting sort in the shape of the memory system, by the bin sort's own partition kernels (vsc_sort.hip; rounds 1-3
    // used rocPRIM's radix sort here): one 8-byte record (key << 32 | site index) per site, level 1 on the top 8 of the 17
    // key bits, level 2 on the other 9 inside every level-1 bin (256 segments) - each level one read and one write of the
    // records, a tile's records of a bin leaving as one contiguous piece.  The order inside a (bucket, class, strand) group is
    // whatever the tiles' reservations give: nothing depends on it.  The level-2 histogram IS the table of group starts.
    constexpr unsigned kKeyBits = 2 * kSegBases + 3, kBits1 = 8, kBits2 = kKeyBits - kBits1;
    constexpr uint32_t kBins1 = 1u << kBits1, kBins2 = 1u << kBits2;
    step(ctx->keys_a.ensure(std::max<uint64_t>(S, 1) * sizeof(uint64_t)));
    step(ctx->keys_b.ensure(std::max<uint64_t>(S, 1) * sizeof(uint64_t)));
    step(full.ensure(std::max<uint64_t>(3 * S, 1) * sizeof(uint4)));
    step(ctx->sort_tabs.ensure(3 * (size_t)(kBins1 * kBins2) * sizeof(uint32_t)));
    const size_t seg_bytes = (kBins1 * sizeof(SortSeg) + 255) / 256 * 256;
    step(ctx->sort_segs.ensure(seg_bytes + (kBins1 + 1) * sizeof(uint32_t)));
    uint4 *const sites16 = (uint4 *)full.p;
    std::vector<uint32_t> bs(kKeys + 1, 0);  // bs[8 b + 2 class + strand] = first site of that group (over all three tables)
    ht.lap("index: order buffers");
    for (int s = 0; s < kSegments && e == hipSuccess && S > 0; ++s) {
        uint64_t *ra = (uint64_t *)ctx->keys_a.p, *rb = (uint64_t *)ctx->keys_b.p;
        SortSeg *d_segs = (SortSeg *)ctx->sort_segs.p;
        uint32_t *d_tile0 = (uint32_t *)((char *)ctx->sort_segs.p + seg_bytes);
        uint32_t *tabs = (uint32_t *)ctx->sort_tabs.p;
        step(launch_seed_keys((const uint4 *)rec16.p, S, s, pam_codes, a.n_pam, ra, st));
        // level 1: one segment, 256 bins
        std::vector<SortSeg> &segs = ctx->host_segs;
        std::vector<uint32_t> &tile0 = ctx->host_tile0;
        const uint32_t tiles = (uint32_t)((S + kSortTile - 1) / kSortTile);
        segs.assign(1, SortSeg{0, 0, 0, (uint32_t)S, 0});
        tile0.assign({0u, tiles});
        SortArgs l1{};
        l1.segs = d_segs;
        l1.seg_tile0 = d_tile0;
        l1.n_segs = 1;
        l1.n_tiles = tiles;
        l1.in = ra;
        l1.out = rb;
        l1.hist = tabs;
        l1.cursor = tabs + kBins1;
        l1.bin_start = tabs + 2 * kBins1;
        l1.bin_bits = kBits1;
        l1.bin_shift = 32 + kBits2;
        if (tiles >= 64) l1.xcd_tiles = (tiles + 7) / 8;
        step(hipMemcpyAsync(d_segs, segs.data(), sizeof(SortSeg), hipMemcpyHostToDevice, st));
        step(hipMemcpyAsync(d_tile0, tile0.data(), 2 * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        step(hipMemsetAsync(l1.hist, 0, kBins1 * sizeof(uint32_t), st));
        if (e == hipSuccess) step(launch_bin_hist(l1, st));
        if (e == hipSuccess) step(launch_bin_scan(l1, st));
        if (e == hipSuccess) step(launch_bin_partition(l1, st));
        std::vector<uint32_t> h1(kBins1);
        step(hipMemcpyAsync(h1.data(), l1.hist, kBins1 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        step(hipStreamSynchronize(st));
        if (e != hipSuccess) break;
        // level 2: every level-1 bin is a segment of its own, 512 bins each
        segs.assign(kBins1, SortSeg{});
        tile0.assign(kBins1 + 1, 0);
        uint64_t at = 0, t2 = 0;
        for (uint32_t i = 0; i < kBins1; ++i) {
            segs[i] = SortSeg{at, at, 0, h1[i], 0};
            tile0[i] = (uint32_t)t2;
            t2 += (h1[i] + kSortTile - 1) / kSortTile;
            at += h1[i];
        }
        tile0[kBins1] = (uint32_t)t2;
        SortArgs l2{};
        l2.segs = d_segs;
        l2.seg_tile0 = d_tile0;
        l2.n_segs = kBins1;
        l2.n_tiles = (uint32_t)t2;
        l2.in = rb;
        l2.out = ra;
        l2.hist = tabs;
        l2.cursor = tabs + kBins1 * kBins2;
        l2.bin_start = tabs + 2 * kBins1 * kBins2;
        l2.bin_bits = kBits2;
        l2.bin_shift = 32;
        if (t2 >= 64) l2.xcd_tiles = (uint32_t)((t2 + 7) / 8);
        step(hipMemcpyAsync(d_segs, segs.data(), kBins1 * sizeof(SortSeg), hipMemcpyHostToDevice, st));
        step(hipMemcpyAsync(d_tile0, tile0.data(), (kBins1 + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        step(hipMemsetAsync(l2.hist, 0, (size_t)kBins1 * kBins2 * sizeof(uint32_t), st));
        if (e == hipSuccess) step(launch_bin_hist(l2, st));
        if (e == hipSuccess) step(launch_bin_scan(l2, st));
        if (e == hipSuccess) step(launch_bin_partition(l2, st));
        if (e == hipSuccess) step(launch_seed_gather16((const uint4 *)rec16.p, ra, S, sites16 + (size_t)s * S, st));
        std::vector<uint32_t> h2((size_t)kBins1 * kBins2);
        step(hipMemcpyAsync(h2.data(), l2.hist, h2.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        step(hipStreamSynchronize(st));  // (the host tables of this table's levels go out of use here)
        if (e != hipSuccess) break;
        uint64_t run = (uint64_t)s * S;  // key = level-1 bin << 9 | level-2 bin: the groups lie in key order
        for (size_t k = 0; k < h2.size(); ++k) {
            bs[(size_t)s * 8 * kBucketsPerSeg + k] = (uint32_t)run;
            run += h2[k];
        }
        if (run != (uint64_t)(s + 1) * S) {
            *why = "index build: the ordered table lost or gained sites";
            e = hipErrorUnknown;
        }
    }
    bs[kKeys] = (uint32_t)(3 * S);
    rec16.release();
    ht.lap("index: three tables ordered");
    step(hipMalloc((void **)&g->d_ix_sites, std::max<uint64_t>(3 * S, 1) * sizeof(uint2)));
    step(hipMalloc((void **)&g->d_ix_edge, edge_words * sizeof(uint32_t)));
    if (e == hipSuccess) step(hipMemsetAsync(g->d_ix_edge, 0, edge_words * sizeof(uint32_t), st));
    if (e == hipSuccess) {
        step(launch_seed_compact(sites16, std::max<uint64_t>(S, 1), 3 * S, g->d_ix_sites, g->d_ix_edge, st));
        step(hipStreamSynchronize(st));
    }
    if (e != hipSuccess) {
        release();
        free_index(g);
        return e;
    }
    ht.lap("index: site records");
    // chunks: at most kSlicedChunk sites of one bucket and class each; the sites of a chunk also exist bit-sliced, in
    // blocks of 32, from block `vfirst` on.  Word z: bucket | rank in the chunk of its first '-' site << 16 | class << 29
    // (| edge flag << 28, set on the device below)
    const uint64_t chunk_sites = kSlicedChunk;
    std::vector<uint32_t> ctab;  // {first site, site count, z, first vertical block} per chunk
    ctab.reserve(4 * (3 * S / chunk_sites + (size_t)kSeedClasses * kBuckets));
    uint64_t n_blocks = 0;
    for (uint32_t b = 0; b < (uint32_t)kBuckets; ++b) {
        for (uint32_t c = 0; c < 4; ++c) {
            const size_t k = 8 * (size_t)b + 2 * c;
            const uint64_t minus = bs[k + 1];
            for (uint64_t p = bs[k]; p < bs[k + 2]; p += chunk_sites) {
                const uint64_t count = std::min<uint64_t>(chunk_sites, bs[k + 2] - p);
                const uint64_t minus_from = minus <= p ? 0 : std::min<uint64_t>(minus - p, count);
                ctab.push_back((uint32_t)p);
                ctab.push_back((uint32_t)count);
                ctab.push_back(b | (uint32_t)(minus_from << kChunkMinusShift) | (c << kChunkClassShift));
                ctab.push_back((uint32_t)n_blocks);
                n_blocks += (count + kSlicedSites - 1) / kSlicedSites;
            }
        }
    }
    g->ix_chunks = (uint32_t)(ctab.size() / 4);
    class_counts(g, ctab);
    const size_t cb = std::max<size_t>(ctab.size(), 4) * sizeof(uint32_t);
    const size_t vb = std::max<uint64_t>(n_blocks, 1) * kVertWords * sizeof(uint32_t);
    step(hipMalloc((void **)&g->d_ix_chunk_tab, cb));
    if (e == hipSuccess && !ctab.empty())
        step(hipMemcpyAsync(g->d_ix_chunk_tab, ctab.data(), ctab.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    step(hipMalloc((void **)&g->d_ix_vert, vb));
    if (e == hipSuccess) step(launch_seed_transpose(sites16, g->d_ix_chunk_tab, g->ix_chunks, g->d_ix_vert, st));
    if (e == hipSuccess) step(launch_seed_chunk_flags(g->d_ix_chunk_tab, g->ix_chunks, g->d_ix_edge, st));
    step(hipEventRecord(ctx->ev[6], st));
    step(hipStreamSynchronize(st));
    ht.lap("index: chunk table, bit-sliced blocks");
    release();
    if (e != hipSuccess) {
        free_index(g);
        return e;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]);
    g->index_ms = ms;
    g->index_sites = S;
    g->sites = S;
    g->has_index = true;
    g->index_has_extra_pam = a.n_pam > 2;
    if (a.n_pam > 2) {
        g->index_extra_pam[0] = params->extra_pam[0];
        g->index_extra_pam[1] = params->extra_pam[1];
    }
    g->ix_vert_bytes = vb;
    g->ix_edge_words = edge_words;
    g->index_bytes = 3 * S * sizeof(uint2) + edge_words * sizeof(uint32_t) + cb + vb;
    return hipSuccess;
}

// ---- seed index on disk (vsc_genome_index_save / _load) ---------------------------------------------
// One header, then the four device arrays as they lie in HBM.  The header pins what the arrays depend on: the
// library's layout constants, the PAM set and a fingerprint of the genome (size, contig table, the first plane words).
struct IndexFileHeader {
    char magic[8];  // "VSCSEED\0"
    uint32_t abi, seg_bases, sliced_chunk, sliced_sites;
    uint64_t sites;  // S
    uint64_t own_words, vert_bytes, edge_words, fingerprint;
    uint32_t chunks, n_contigs;
    uint8_t has_extra_pam;
    char extra_pam[2];
    uint8_t layout;  // kIndexLayout
    uint8_t pad[4];
};
static_assert(sizeof(IndexFileHeader) == 80, "index file header layout");
constexpr char kIndexMagic[8] = {'V', 'S', 'C', 'S', 'E', 'E', 'D', 0};
constexpr size_t kIndexIoChunk = 64u << 20;

// Every word of all three planes of the shard (a device-side reduction: under 1 ms at 3 Gbp) + the contig table
// (offsets and ends) + the shard's place in the genome.  A genome of the same size and layout that differs in one
// base, or only in what is masked as N, gets another fingerprint.
hipError_t genome_fingerprint(vsc_ctx *ctx, const vsc_genome *g, uint64_t *out)
{
    VSC_TRY(ctx->counters.ensure(kCounterWords * sizeof(unsigned long long)));
    unsigned long long *d_sum = (unsigned long long *)ctx->counters.p;
    VSC_TRY(hipMemsetAsync(d_sum, 0, sizeof *d_sum, ctx->stream));
    const uint64_t words = std::min<uint64_t>(g->own_words + 1, g->dev_words);  // own words + the halo word
    VSC_TRY(launch_plane_hash(g->d_hi, g->d_lo, g->d_nm, words, d_sum, ctx->stream));
    unsigned long long sum = 0;
    VSC_TRY(hipMemcpyAsync(&sum, d_sum, sizeof sum, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint32_t> tab(2 * (size_t)g->n_contigs);
    if (g->n_contigs) {
        VSC_TRY(hipMemcpyAsync(tab.data(), g->d_contig_off, (size_t)g->n_contigs * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        VSC_TRY(hipMemcpyAsync(tab.data() + g->n_contigs, g->d_contig_end, (size_t)g->n_contigs * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    VSC_TRY(hipStreamSynchronize(ctx->stream));
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a over the rest
    auto eat = [&](uint64_t v) {
        for (int i = 0; i < 8; ++i) h = (h ^ ((v >> (8 * i)) & 0xFF)) * 0x100000001b3ull;
    };
    eat(sum);
    eat(g->own_words);
    eat(g->first_word);
    for (uint32_t w : tab) eat(w);
    *out = h;
    return hipSuccess;
}

// What seed_sliced_kernel reads through the chunk table must lie inside the arrays: a damaged file whose length
// happens to match its header must not become out-of-bounds device reads.
bool chunk_table_ok(const std::vector<uint32_t> &ctab, uint64_t S, uint64_t vert_bytes, std::string *why)
{
    const uint64_t n_blocks = vert_bytes / (kVertWords * sizeof(uint32_t));
    uint64_t expect_site = 0, expect_block = 0;
    uint32_t last_group = 0;  // bucket << 2 | class: the chunks lie in this order
    for (size_t c = 0; c * 4 < ctab.size(); ++c) {
        const uint64_t first = ctab[4 * c], count = ctab[4 * c + 1], vfirst = ctab[4 * c + 3];
        const uint32_t z = ctab[4 * c + 2], bucket = z & kChunkBucketMask, minus_from = (z >> kChunkMinusShift) & 0xFFFu;
        const uint32_t cls = z >> kChunkClassShift;
        const uint64_t blocks = (count + kSlicedSites - 1) / kSlicedSites;
        const char *bad = nullptr;
        if (count == 0 || count > (uint64_t)kSlicedChunk) bad = "site count";
        else if (first != expect_site || first + count > 3 * S) bad = "first site";
        else if (vfirst != expect_block || vfirst + blocks > n_blocks) bad = "first block";
        else if (bucket >= (uint32_t)kBuckets || cls >= (uint32_t)kSeedClasses || (bucket << 2 | cls) < last_group) bad = "bucket";
        else if (minus_from > count) bad = "strand boundary";
        if (bad) {
            *why = "chunk " + std::to_string(c) + ": bad " + bad;
            return false;
        }
        expect_site = first + count;
        expect_block = vfirst + blocks;
        last_group = bucket << 2 | cls;
    }
    if (expect_site != 3 * S) {
        *why = "the chunks do not cover the site table";
        return false;
    }
    return true;
}

// sites and chunks per PAM class (what the search's cost model wants to know about the index)
void class_counts(vsc_genome *g, const std::vector<uint32_t> &ctab)
{
    for (int c = 0; c < 4; ++c) g->ix_class_sites[c] = g->ix_class_chunks[c] = 0;
    for (size_t c = 0; c * 4 < ctab.size(); ++c) {
        const uint32_t cls = ctab[4 * c + 2] >> kChunkClassShift;
        g->ix_class_sites[cls] += ctab[4 * c + 1];
        g->ix_class_chunks[cls] += 1;
    }
    for (int c = 0; c < 4; ++c) g->ix_class_sites[c] /= kSegments;  // every site is filed once per table
}

struct FileCloser {
    std::FILE *f;
    ~FileCloser() { if (f) std::fclose(f); }
};

// device array <-> file, through one host buffer
hipError_t index_io(std::FILE *f, void *dev, uint64_t bytes, bool to_file, std::vector<char> &host, bool *io_ok)
{
    for (uint64_t off = 0; off < bytes; off += kIndexIoChunk) {
        const size_t n = (size_t)std::min<uint64_t>(kIndexIoChunk, bytes - off);
        if (to_file) {
            VSC_TRY(hipMemcpy(host.data(), (const char *)dev + off, n, hipMemcpyDeviceToHost));
            if (std::fwrite(host.data(), 1, n, f) != n) { *io_ok = false; return hipSuccess; }
        } else {
            if (std::fread(host.data(), 1, n, f) != n) { *io_ok = false; return hipSuccess; }
            VSC_TRY(hipMemcpy((char *)dev + off, host.data(), n, hipMemcpyHostToDevice));
        }
    }
    return hipSuccess;
}

}  // namespace

int vsc_genome_build_index(vsc_ctx *ctx, vsc_genome *genome, const vsc_search_params *params)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx || !genome || genome->ctx != ctx) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (index_matches(genome, params)) return VSC_OK;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    std::string why;
    hipError_t e = build_index(ctx, genome, params, &why);
    if (e != hipSuccess) {
        if (!why.empty()) return fail(ctx, VSC_ERR_RANGE, ("vsc_genome_build_index: " + why).c_str());
        return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, "vsc_genome_build_index", e);
    }
    ctx->timing.index_ms = genome->index_ms;
    return VSC_OK;
    });
}

int vsc_genome_index_save(vsc_ctx *ctx, const vsc_genome *genome, const char *path)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx || !genome || genome->ctx != ctx || !path) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (!genome->has_index) return fail(ctx, VSC_ERR_INVALID, "vsc_genome_index_save: the genome has no seed index (vsc_genome_build_index)");
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    IndexFileHeader h{};
    std::memcpy(h.magic, kIndexMagic, sizeof h.magic);
    h.abi = VSC_ABI_VERSION;
    h.seg_bases = kSegBases;
    h.sliced_chunk = kSlicedChunk;
    h.sliced_sites = kSlicedSites;
    h.layout = (uint8_t)kIndexLayout;
    h.sites = genome->index_sites;
    h.own_words = genome->own_words;
    h.vert_bytes = genome->ix_vert_bytes;
    h.edge_words = genome->ix_edge_words;
    h.chunks = genome->ix_chunks;
    h.n_contigs = genome->n_contigs;
    h.has_extra_pam = genome->index_has_extra_pam;
    h.extra_pam[0] = genome->index_extra_pam[0];
    h.extra_pam[1] = genome->index_extra_pam[1];
    VSC_HIP(ctx, genome_fingerprint(ctx, genome, &h.fingerprint));
    FileCloser fc{std::fopen(path, "wb")};
    if (!fc.f) return fail(ctx, VSC_ERR_INVALID, (std::string("vsc_genome_index_save: cannot write ") + path).c_str());
    bool ok = std::fwrite(&h, sizeof h, 1, fc.f) == 1;
    std::vector<char> host(kIndexIoChunk);
    const uint64_t S = genome->index_sites;
    if (ok) VSC_HIP(ctx, index_io(fc.f, genome->d_ix_sites, 3 * S * sizeof(uint2), true, host, &ok));
    if (ok) VSC_HIP(ctx, index_io(fc.f, genome->d_ix_edge, h.edge_words * sizeof(uint32_t), true, host, &ok));
    if (ok) VSC_HIP(ctx, index_io(fc.f, genome->d_ix_chunk_tab, (uint64_t)h.chunks * sizeof(uint4), true, host, &ok));
    if (ok) VSC_HIP(ctx, index_io(fc.f, genome->d_ix_vert, h.vert_bytes, true, host, &ok));
    if (ok) ok = std::fflush(fc.f) == 0;
    if (!ok) return fail(ctx, VSC_ERR_DEVICE, (std::string("vsc_genome_index_save: write to ") + path + " failed").c_str());
    return VSC_OK;
    });
}

int vsc_genome_index_load(vsc_ctx *ctx, vsc_genome *genome, const char *path)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx || !genome || genome->ctx != ctx || !path) return VSC_ERR_INVALID;
    ctx->err.clear();
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    FileCloser fc{std::fopen(path, "rb")};
    if (!fc.f) return fail(ctx, VSC_ERR_INVALID, (std::string("vsc_genome_index_load: cannot read ") + path).c_str());
    IndexFileHeader h{};
    if (std::fread(&h, sizeof h, 1, fc.f) != 1 || std::memcmp(h.magic, kIndexMagic, sizeof h.magic) != 0)
        return fail(ctx, VSC_ERR_INVALID, (std::string("vsc_genome_index_load: ") + path + " is not a seed index file").c_str());
    if (h.abi != VSC_ABI_VERSION || h.seg_bases != (uint32_t)kSegBases || h.sliced_chunk != (uint32_t)kSlicedChunk ||
        h.sliced_sites != (uint32_t)kSlicedSites || h.layout != (uint8_t)kIndexLayout)
        return fail(ctx, VSC_ERR_INVALID, "vsc_genome_index_load: the file was written by another version of the library");
    uint64_t fp = 0;
    VSC_HIP(ctx, genome_fingerprint(ctx, genome, &fp));
    if (h.own_words != genome->own_words || h.n_contigs != genome->n_contigs || h.fingerprint != fp)
        return fail(ctx, VSC_ERR_INVALID, "vsc_genome_index_load: the file belongs to another genome");
    const uint64_t S = h.sites;
    if (3 * S >= (1ull << 32) || h.edge_words != (3 * S + 31) / 32 + 1 || h.vert_bytes % (kVertWords * sizeof(uint32_t)) != 0)
        return fail(ctx, VSC_ERR_INVALID, "vsc_genome_index_load: inconsistent header");
    {
        // the arrays the header announces must be what the file holds (a cut or padded file is refused before
        // anything is allocated)
        const long at = std::ftell(fc.f);
        const uint64_t want = (uint64_t)sizeof h + 3 * S * sizeof(uint2) + h.edge_words * sizeof(uint32_t) +
                              (uint64_t)h.chunks * sizeof(uint4) + h.vert_bytes;
        if (at < 0 || std::fseek(fc.f, 0, SEEK_END) != 0) return fail(ctx, VSC_ERR_INVALID, "vsc_genome_index_load: cannot seek in the file");
        const long size = std::ftell(fc.f);
        if (size < 0 || (uint64_t)size != want || std::fseek(fc.f, at, SEEK_SET) != 0)
            return fail(ctx, VSC_ERR_INVALID, (std::string("vsc_genome_index_load: ") + path + " is truncated or does not match its header").c_str());
    }
    free_index(genome);
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    const uint64_t sb = std::max<uint64_t>(3 * S, 1) * sizeof(uint2), eb = h.edge_words * sizeof(uint32_t);
    const uint64_t cb = std::max<uint64_t>(h.chunks, 1) * sizeof(uint4), vb = std::max<uint64_t>(h.vert_bytes, kVertWords * sizeof(uint32_t));
    step(hipMalloc((void **)&genome->d_ix_sites, sb));
    step(hipMalloc((void **)&genome->d_ix_edge, eb));
    step(hipMalloc((void **)&genome->d_ix_chunk_tab, cb));
    step(hipMalloc((void **)&genome->d_ix_vert, vb));
    bool ok = true;
    std::vector<char> host(kIndexIoChunk);
    if (e == hipSuccess) step(index_io(fc.f, genome->d_ix_sites, 3 * S * sizeof(uint2), false, host, &ok));
    if (e == hipSuccess && ok) step(index_io(fc.f, genome->d_ix_edge, eb, false, host, &ok));
    if (e == hipSuccess && ok) step(index_io(fc.f, genome->d_ix_chunk_tab, (uint64_t)h.chunks * sizeof(uint4), false, host, &ok));
    if (e == hipSuccess && ok) step(index_io(fc.f, genome->d_ix_vert, h.vert_bytes, false, host, &ok));
    if (e != hipSuccess || !ok) {
        free_index(genome);
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, "vsc_genome_index_load", e);
        return fail(ctx, VSC_ERR_INVALID, (std::string("vsc_genome_index_load: ") + path + " is truncated").c_str());
    }
    {
        std::vector<uint32_t> ctab((size_t)h.chunks * 4);
        if (h.chunks) VSC_HIP(ctx, hipMemcpy(ctab.data(), genome->d_ix_chunk_tab, ctab.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::string why;
        if (!chunk_table_ok(ctab, S, h.vert_bytes, &why)) {
            free_index(genome);
            return fail(ctx, VSC_ERR_INVALID, (std::string("vsc_genome_index_load: ") + path + " is damaged (" + why + ")").c_str());
        }
        class_counts(genome, ctab);
    }
    genome->ix_chunks = h.chunks;
    genome->ix_vert_bytes = vb;
    genome->ix_edge_words = h.edge_words;
    genome->index_sites = S;
    genome->sites = S;
    genome->has_index = true;
    genome->index_has_extra_pam = h.has_extra_pam;
    genome->index_extra_pam[0] = h.extra_pam[0];
    genome->index_extra_pam[1] = h.extra_pam[1];
    genome->index_bytes = sb + eb + cb + vb;
    genome->index_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->timing.index_ms = genome->index_ms;
    return VSC_OK;
    });
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// search
// ------------------------------------------------------------------------------------------------
namespace {

unsigned ceil_log2(uint64_t x)
{
    unsigned b = 0;
    while (b < 63 && (1ull << b) < x) ++b;
    return b;
}

// what one bin sort did (goes into vsc_timing)
// A search that keeps the sites' bases (vsc_search_stream_rows): every record travels with a 32-bit side word (the site's lo
// plane; parallel arrays beside the two record buffers) and the finalize kernel writes the hits' packed feature rows.
struct SortRows {
    uint32_t *side_src = nullptr;     // side words beside `src`
    DeviceBuf *side_other = nullptr;  // grows to the size of `other` (in records) x 4 bytes
    const uint2 *guides = nullptr;    // planes of the pass's reads
    uint32_t guide_first = 0;         // read index of guides[0]
    uint4 *rows = nullptr;            // row of result record i at rows + (i - rows_first)
    uint64_t rows_first = 0;
};

struct SortInfo {
    unsigned levels = 0;     // partition levels run (0: every region fitted the finalize kernel as it was)
    unsigned bin_bits = 0;   // key bits of the first partition level
    uint64_t bytes = 0;      // bytes the sort kernels moved: 8 per record read or written, 16 per result record
    unsigned slot_fallbacks = 0;  // slot-mode levels that overflowed and ran again with the histogram
};

// Orders the packed records of `segs` (each segment independently, ascending) and writes the vsc_hit
// records: hist -> scan -> partition per level, then finalize; bins too large for the finalize kernel come
// back as the segments of the next level (vsc_sort.hip).  `src` holds the segments' records, `other` is a
// buffer of the same size; `key_bits` = significant bits of record >> kRecPosShift inside a segment.
// pair_keys != null: level 0 of the streaming scan - one segment of (key, value) pairs, partitioned by region
// into `other` and packed on the way; the regions then are the segments.
// slots != null (the seed search's level 1): the first partition level runs in SLOT MODE when *slots allows it -
// no histogram pass: every bin owns a fixed slot of `other` (grown to n_bins x slot capacity through `other_buf`), the
// partition reserves room with the bin cursors alone; a bin that outgrows its slot (repeats) makes the level run again
// with the histogram and clears *slots, so that later searches of this genome and budget go the exact way at once.
hipError_t bin_sort(vsc_ctx *ctx, const vsc_genome *genome, std::vector<SortSeg> segs, uint64_t *src, uint64_t *other,
                    unsigned key_bits, unsigned pos_pad, uint32_t pos_base, vsc_hit *out, hipEvent_t ev_sorted, SortInfo *info,
                    DeviceBuf *other_buf = nullptr, bool *slots = nullptr, const SortRows *rows = nullptr)
{
    uint32_t *side_src = rows ? rows->side_src : nullptr, *side_other = nullptr;
    // (key_bits counts the meaningful bits: the pos_pad zero bits at the bottom of every position field are not among them)
    hipStream_t st = ctx->stream;
    unsigned rem = key_bits;  // key bits no partition level has used yet
    bool sorted_marked = false;
    // test hooks (varscot_hip_debug.h): a smaller bin capacity / fewer bits per level make the partition levels
    // and the oversize path run on inputs of a few thousand records
    const vsc_debug_params &dbg = ctx->dbg;
    uint64_t sort_cap = kSortCap;
    unsigned max_bits = kSortMaxBinBits;
    if (dbg.sort_cap) sort_cap = std::min<uint64_t>(kSortCap, std::max<uint32_t>(16, dbg.sort_cap));
    if (dbg.sort_max_bits) max_bits = std::min<unsigned>(kSortMaxBinBits, std::max<uint32_t>(1, dbg.sort_max_bits));
    // a slot holds a quarter more than the finalize kernel orders in LDS: bins between the two go to another level
    // from their slot, bins beyond make the level fall back
    uint64_t slot_cap = dbg.sort_slot_cap ? dbg.sort_slot_cap : (sort_cap + sort_cap / 4 + 15) / 16 * 16;
    slot_cap = std::max<uint64_t>(slot_cap, 16);
    for (unsigned level = 1; !segs.empty(); ++level) {
        if (level > 48) return hipErrorUnknown;  // cannot happen: every level consumes key bits, keys are unique
        uint64_t n_max = 0, n_all = 0;
        for (const SortSeg &s : segs) {
            n_max = std::max<uint64_t>(n_max, s.n_in);
            n_all += s.n_in;
        }
        unsigned bits = 0;
        if (n_max > sort_cap) {
            if (rem == 0) return hipErrorUnknown;
            const uint64_t want = std::max<uint64_t>(1, sort_cap * 7 / 10);  // average bin: 70 % of what the finalize kernel holds
            bits = std::min<unsigned>({std::max(1u, ceil_log2((n_max + want - 1) / want)), max_bits, rem});
            while (bits > 1 && ((uint64_t)segs.size() << bits) > (1ull << 22)) --bits;  // bounded bin tables
        }
        const size_t n_segs = segs.size();
        const size_t n_bins = n_segs << bits;
        bool use_slots = level == 1 && bits && slots && other_buf && (dbg.sort_optimistic == 1 || (dbg.sort_optimistic != 0 && *slots));
        if (level == 1 && bits && other_buf) {
            // the level's destination is sized here, once, for the layout that will be used: fixed slots (n_bins x slot
            // capacity: 1.8 - 3.6 x the records with average bins at 70 % of the finalize kernel's capacity) when they are
            // allowed and fit the device beside everything else, the compact layout otherwise
            uint64_t span = 1;
            for (const SortSeg &sg : segs) span = std::max<uint64_t>(span, sg.out_off + sg.n_in);
            const uint64_t slot_records = (uint64_t)n_bins * slot_cap;
            if (use_slots && dbg.sort_optimistic != 1) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
                const uint64_t room = (uint64_t)free_b + other_buf->cap;  // (the buffer's present allocation is given back first)
                // (a search that keeps the sites' bases has a side word beside every slot)
                const uint64_t slot_bytes = slot_records * (sizeof(uint64_t) + (rows ? sizeof(uint32_t) : 0));
                if (slot_records > 4 * n_all + (1ull << 20) || slot_bytes + (1ull << 30) > room + (rows ? rows->side_other->cap : 0)) {
                    use_slots = false;
                    *slots = false;  // remembered per genome and budget: later searches do not ask again
                }
            }
            // (never less than the compact layout: a slot partition that overflows runs again the exact way, into the same buffer)
            hipError_t ge = other_buf->ensure((size_t)(use_slots ? std::max(slot_records, span) : span) * sizeof(uint64_t));
            if (ge == hipErrorOutOfMemory && use_slots) {
                (void)hipGetLastError();
                use_slots = false;
                *slots = false;
                ge = other_buf->ensure((size_t)span * sizeof(uint64_t));
            }
            if (ge != hipSuccess) return ge;
            other = (uint64_t *)other_buf->p;
            if (rows) {  // the side words' second buffer: as many words as `other` has records
                VSC_TRY(rows->side_other->ensure(other_buf->cap / sizeof(uint64_t) * sizeof(uint32_t)));
                side_other = (uint32_t *)rows->side_other->p;
            }
        }
        std::vector<uint32_t> &tile0 = ctx->host_tile0;
        tile0.assign(n_segs + 1, 0);
        uint64_t tiles = 0;
        for (size_t i = 0; i < n_segs; ++i) {
            tile0[i] = (uint32_t)tiles;
            tiles += (segs[i].n_in + kSortTile - 1) / kSortTile;
        }
        if (tiles >= (1ull << 31)) return hipErrorInvalidValue;
        tile0[n_segs] = (uint32_t)tiles;
        const size_t seg_bytes = (n_segs * sizeof(SortSeg) + 255) / 256 * 256;
        VSC_TRY(ctx->sort_segs.ensure(seg_bytes + tile0.size() * sizeof(uint32_t)));
        SortSeg *d_segs = (SortSeg *)ctx->sort_segs.p;
        uint32_t *d_tile0 = (uint32_t *)((char *)ctx->sort_segs.p + seg_bytes);
        ctx->host_segs = segs;  // (the copy the asynchronous upload reads from)
        VSC_TRY(hipMemcpyAsync(d_segs, ctx->host_segs.data(), n_segs * sizeof(SortSeg), hipMemcpyHostToDevice, st));
        VSC_TRY(hipMemcpyAsync(d_tile0, tile0.data(), tile0.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        VSC_TRY(ctx->sort_over.ensure(256 + n_bins * sizeof(SortSeg)));
        uint32_t *d_n_over = (uint32_t *)ctx->sort_over.p;  // [0] listed bins, [1] the finalize kernel's bin cursor, [2] slot overflow
        if (bits) VSC_TRY(ctx->sort_tabs.ensure(3 * n_bins * sizeof(uint32_t)));
        uint32_t flags[3] = {0, 0, 0};
        for (;;) {  // once; twice when the slot partition overflowed
            if (dbg.sort_debug)
                std::fprintf(stderr, "[vsc sort] level %u: %zu segments, %llu records, largest %llu, %u bits (of %u left), cap %llu%s\n", level,
                             n_segs, (unsigned long long)n_all, (unsigned long long)n_max, bits, rem, (unsigned long long)sort_cap,
                             use_slots ? ", slot mode" : "");
            VSC_TRY(hipMemsetAsync(d_n_over, 0, 3 * sizeof(uint32_t), st));
            FinArgs f{};
            f.segs = d_segs;
            f.n_segs = (uint32_t)n_segs;
            f.src = src;
            f.side_src = side_src;
            if (rows) {
                f.guides = rows->guides;
                f.guide_first = rows->guide_first;
                f.rows = rows->rows;
                f.rows_first = rows->rows_first;
            }
            if (bits) {
                SortArgs a{};
                a.segs = d_segs;
                a.seg_tile0 = d_tile0;
                a.n_segs = (uint32_t)n_segs;
                a.n_tiles = (uint32_t)tiles;
                a.in = src;
                a.out = other;
                a.side_in = rows ? side_src : nullptr;
                a.side_out = rows ? side_other : nullptr;
                a.hist = (uint32_t *)ctx->sort_tabs.p;
                a.cursor = a.hist + n_bins;
                a.bin_start = a.cursor + n_bins;
                a.bin_bits = bits;
                a.bin_shift = kRecPosShift + pos_pad + rem - bits;
                if (dbg.sort_xcd != 0 && tiles >= 64) a.xcd_tiles = (uint32_t)((tiles + 7) / 8);
                if (use_slots) {
                    // partition first (cursors from zero), then the bin starts of the RESULT from the cursors
                    a.hist = a.cursor;
                    a.slot_cap = (uint32_t)slot_cap;
                    a.overflow = d_n_over + 2;
                    VSC_TRY(hipMemsetAsync(a.cursor, 0, n_bins * sizeof(uint32_t), st));
                    VSC_TRY(launch_bin_partition(a, st));
                    VSC_TRY(launch_bin_scan(a, st));
                } else {
                    VSC_TRY(hipMemsetAsync(a.hist, 0, n_bins * sizeof(uint32_t), st));
                    VSC_TRY(launch_bin_hist(a, st));
                    VSC_TRY(launch_bin_scan(a, st));
                    if (dbg.sort_debug >= 2) {
                        // recount every bin on the host and compare with the device histogram
                        std::vector<uint32_t> h(n_bins);
                        VSC_TRY(hipMemcpyAsync(h.data(), a.hist, n_bins * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                        VSC_TRY(hipStreamSynchronize(st));
                        for (size_t i = 0; i < n_segs; ++i) {
                            std::vector<uint64_t> recs(segs[i].n_in);
                            VSC_TRY(hipMemcpy(recs.data(), src + segs[i].in_off, recs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                            std::vector<uint32_t> want((size_t)1 << bits, 0);
                            for (uint64_t r : recs) {
                                if (r >> 63) continue;
                                want[(r >> a.bin_shift) & (((uint64_t)1 << bits) - 1)]++;
                            }
                            for (size_t b = 0; b < want.size(); ++b)
                                if (want[b] != h[(i << bits) + b])
                                    std::fprintf(stderr, "[vsc sort]   seg %zu (in_off %llu n %u) bin %zu: device %u host %u\n", i,
                                                 (unsigned long long)segs[i].in_off, segs[i].n_in, b, h[(i << bits) + b], want[b]);
                        }
                    }
                    VSC_TRY(launch_bin_partition(a, st));
                }
                f.src = other;
                f.side_src = side_other;
                f.hist = a.hist;
                f.bin_start = a.bin_start;
                f.bin_bits = bits;
                f.slot_cap = a.slot_cap;
                f.overflow = a.overflow;
            }
            if (!sorted_marked && ev_sorted) {
                VSC_TRY(hipEventRecord(ev_sorted, st));
                sorted_marked = true;
            }
            const unsigned rem_after = rem - bits;
            f.sub_bits = std::min<unsigned>(rows ? kSortSubBitsRows : kSortSubBits, rem_after);
            f.sub_shift = kRecPosShift + pos_pad + rem_after - f.sub_bits;
            f.pos_pad = pos_pad;
            f.pos_base = pos_base;
            f.low_bits = rem_after - f.sub_bits;
            f.over = (SortSeg *)((char *)ctx->sort_over.p + 256);
            f.over_cap = (uint32_t)std::min<size_t>(n_bins, 0xFFFFFFFFu);
            f.n_over = d_n_over;
            f.cursor = d_n_over + 1;
            f.cap = (uint32_t)sort_cap;
            f.contig_off = genome->d_contig_off;
            f.n_contigs = genome->n_contigs;
            f.out = out;
            VSC_TRY(launch_bin_finalize(f, 2 * ctx->n_cus, st));  // two workgroups fit a CU (LDS)
            if (bits) {
                VSC_TRY(hipMemcpyAsync(flags, d_n_over, sizeof flags, hipMemcpyDeviceToHost, st));
                VSC_TRY(hipStreamSynchronize(st));
                if (use_slots && flags[2]) {
                    // a bin outgrew its slot: the source is untouched, nothing was finalized (the finalize kernel left at
                    // once) - the same level again, exactly.  What the failed attempt moved: the partition's read + write
                    use_slots = false;
                    *slots = false;
                    if (info) {
                        info->slot_fallbacks++;
                        info->bytes += 16 * n_all;
                    }
                    continue;
                }
            }
            if (info && bits) {
                if (info->levels == 0) info->bin_bits = bits;
                info->levels++;
                info->bytes += (use_slots ? 16 : 24) * n_all;  // (histogram read,) partition read + write
            }
            if (info) info->bytes += 24 * n_all;  // finalize: 8-byte read, 16-byte write
            if (info && rows) info->bytes += (bits ? 8 : 0) * n_all + (4 + 12 + 64) * n_all;  // side words moved, read (twice: + the records again), rows written
            break;
        }
        if (!bits) break;
        rem -= bits;
        const uint32_t n_over = flags[0];
        if (n_over == 0) break;
        segs.resize(n_over);
        VSC_TRY(hipMemcpyAsync(segs.data(), (char *)ctx->sort_over.p + 256, (size_t)n_over * sizeof(SortSeg), hipMemcpyDeviceToHost, st));
        VSC_TRY(hipStreamSynchronize(st));
        std::swap(src, other);  // the listed bins are spans of the buffer this level wrote
        std::swap(side_src, side_other);
        if (rows && !side_other) {  // (level 1 did not partition: the first buffer's sibling has no twin yet)
            uint64_t span = 1;
            for (const SortSeg &sg : segs) span = std::max<uint64_t>(span, sg.out_off + sg.n_in);
            VSC_TRY(rows->side_other->ensure(span * sizeof(uint32_t)));
            side_other = (uint32_t *)rows->side_other->p;
        }
    }
    return hipSuccess;
}

// Room for `n` more records behind the `used` records a result already holds (multi-pass searches grow
// their result; `projected` = expected final size, allocated at once when the buffer has to grow).
hipError_t result_room(vsc_ctx *ctx, vsc_hits *hits, uint64_t used, uint64_t n, uint64_t projected)
{
    const uint64_t need = used + n;
    if (hits->storage.p && hits->storage.cap >= need * sizeof(vsc_hit)) return hipSuccess;
    if (used == 0) {
        if (hits->storage.p) {
            ctx->spare_records.push_back(hits->storage);
            hits->storage = DeviceBuf{};
        }
        VSC_TRY(take_records(ctx, hits, std::max(need, projected)));
        return hipSuccess;
    }
    DeviceBuf bigger;
    VSC_TRY(bigger.ensure(std::max(need, projected) * sizeof(vsc_hit)));
    hipError_t e = hipMemcpyAsync(bigger.p, hits->storage.p, used * sizeof(vsc_hit), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        bigger.release();
        return e;
    }
    hits->storage.release();
    hits->storage = bigger;
    hits->d_records = (vsc_hit *)bigger.p;
    return hipSuccess;
}

struct PassResult {
    uint64_t n = 0;  // records this pass appended to the result
};

// One search pass: reads guides[0 .. n_guides) (n_guides <= kMaxPassReads), reported as read indices
// guide_base + i, appended to `hits` behind the `used` records it already holds.  Timings are ADDED to t.
// want_rows (SEED only): the search keeps the sites' bases beside the records and the sort's last stage writes every hit's
// packed feature row into ctx->score_feat (row i of the pass at byte 64 i).
int search_pass(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides, uint32_t guide_base,
                const vsc_search_params *params, int algo, vsc_hits *hits, uint64_t used, uint64_t projected, vsc_timing &t,
                PassResult *res, bool want_rows = false)
{
#define VSC_HIP_H(call)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, #call, e_); \
    } while (0)
    HostTimer ht;
    // reads as (hi, lo) plane pairs, padded to the unroll factor with reads that can never match
    const uint32_t n_pad = (n_guides + kGuideUnroll - 1) / kGuideUnroll * kGuideUnroll;
    std::vector<uint32_t> gp((size_t)(n_pad + kGuideUnroll) * 2, 0xFFFFFFFFu);  // + one prefetch group
    for (uint32_t i = 0; i < n_guides; ++i) guide_planes(guides[i], &gp[2 * (size_t)i], &gp[2 * (size_t)i + 1]);
    VSC_HIP_H(ctx->guides.ensure(gp.size() * sizeof(uint32_t)));
    VSC_HIP_H(ctx->counters.ensure(kCounterWords * sizeof(unsigned long long)));
    VSC_HIP_H(hipEventRecord(ctx->ev[0], ctx->stream));
    VSC_HIP_H(hipMemcpyAsync(ctx->guides.p, gp.data(), gp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));

    const double own_bases = (double)genome->n_tiles * kTileBases;
    const double sites_est = genome->sites ? (double)genome->sites : own_bases / 4;
    uint64_t cap = (uint64_t)(1.5 * sites_est * n_guides * hit_probability(params->max_mismatches)) + (1u << 20);
    const double seen_rate = genome->seen_rate[params->max_mismatches];  // 0: no search with this budget yet
    if (algo == VSC_ALGO_SCAN) cap = std::max<uint64_t>(cap, (uint64_t)(1.2 * seen_rate * n_guides) + 4096);
    unsigned long long cnt[kCntPart + 4 * kParts] = {};
    const int n_parts = (int)((n_guides + kRegionReads - 1) / kRegionReads);  // output regions of 64 reads
    // bits the positions of this shard need, counted from the shard's first position (a shard of the upper part of a
    // genome must not leave the top key bits constant); the records keep them at the top of their 32-bit position field
    const uint32_t pos_base = (uint32_t)(genome->first_word * 32);
    const unsigned pos_bits = std::max(1u, ceil_log2((uint64_t)genome->dev_words * 32));
    const unsigned pos_pad = pos_bits < 32 ? 32 - pos_bits : 0;

    ScanArgs a{};
    SeedArgs sa{};
    uint64_t part_cap = 0;
    int n_groups = 1;
    bool seed_shared = false;
    if (algo == VSC_ALGO_SCAN) {
        fill_pam(a, params);
        fill_genome(a, ctx, genome);
        a.guides = (const uint4 *)ctx->guides.p;
        a.n_guides_padded = n_pad;
        a.max_mm = params->max_mismatches;
        a.k_half = params->max_mismatches / 2;  // bidir_mapping.cpp:129-146
        n_groups = scan_groups(ctx, a.n_tiles);
        t.genome_bytes += (uint64_t)genome->n_tiles * kTileWords * 3 * sizeof(uint32_t);
        VSC_HIP_H(hipEventRecord(ctx->ev[7], ctx->stream));
    } else {
        // ---- per-bucket read lists: a counting sort of the reads' segment neighbourhoods over the buckets ------
        // The cut of the pigeonhole (SeedPlan, vsc_internal.h): segment 0 within k0 substitutions, segment 1 within k1, segment 2
        // within what the site's PAM class leaves of the limit - k0 - k1 - 2.  Hook seed_tight = 0: floor(m / 3) everywhere (the
        // round-3 cut); 1 + k0 + 3 k1: that cut, if it is a valid one.
        const uint32_t m = params->max_mismatches;
        auto nbr = [](int k) { return k < 0 ? 0u : (k == 0 ? 1u : (k == 1 ? 22u : 211u)); };
        SeedPlan plan{};
        plan.max_mm = m;
        {
            ScanArgs pa{};
            vsc_search_params ip{};
            ip.has_extra_pam = genome->index_has_extra_pam;
            ip.extra_pam[0] = genome->index_extra_pam[0];
            ip.extra_pam[1] = genome->index_extra_pam[1];
            fill_pam(pa, &ip);  // the classes of the index, in the order build_index gave them
            plan.n_pam = pa.n_pam;
            plan.pam_codes = pam_code_set(pa);
        }
        plan.tight = ctx->dbg.seed_tight != 0;
        int k2_max = 0;
        if (!plan.tight) {
            plan.k0 = plan.k1 = m / kSegments;
            k2_max = (int)plan.k0;
        } else {
            // reads per (class, what the class leaves them for positions 0..20)
            uint32_t left_reads[kSeedClasses][VSC_MAX_MISMATCHES + 1] = {};
            int left_max = -1;
            for (uint32_t i = 0; i < n_guides; ++i) {
                const uint32_t x = gp[2 * (size_t)i], l = gp[2 * (size_t)i + 1];
                const uint32_t mine = (((x >> 21) & 1u) << 3) | (((l >> 21) & 1u) << 2) | (((x >> 22) & 1u) << 1) | ((l >> 22) & 1u);
                for (uint32_t c = 0; c < plan.n_pam; ++c) {
                    const uint32_t diff = mine ^ ((plan.pam_codes >> (4 * c)) & 15u);
                    const uint32_t spent = ((diff & 12u) ? 1u : 0u) + ((diff & 3u) ? 1u : 0u);
                    if (spent > m) continue;
                    left_reads[c][m - spent]++;
                    left_max = std::max(left_max, (int)(m - spent));
                }
            }
            // Every (k0, k1) in 0..2 that keeps the third threshold within two substitutions is a valid cut; they differ in what
            // they cost.  Fewer buckets per read = fewer comparisons (what a dense search is bound by: c3), but a table whose
            // lists are short still has nearly all of its chunks loaded for a read or two each (what a sparse search is bound by:
            // at 1 000 reads and m = 6 the cut (2, 2, 0) compares 2.6 x the pairs of (1, 1, 2) and loads 14 % fewer blocks).
            // cost = max(block bytes at 4.2 TB/s, chunk visits x 215 SIMD-cycles over all SIMDs at 2 GHz)
            double cost[9];  // [k0 + 3 k1]; < 0: not a valid cut
            double chunks_all = 0, sites_all = 0;
            uint64_t reads_any = 0;
            for (uint32_t c = 0; c < plan.n_pam; ++c) {
                chunks_all += (double)genome->ix_class_chunks[c] / kSegments;
                sites_all += (double)genome->ix_class_sites[c];
                uint64_t reads_c = 0;
                for (int left = 0; left <= (int)m; ++left) reads_c += left_reads[c][left];
                reads_any = std::max(reads_any, reads_c);
            }
            const double block_bytes = (double)kVertWords * 4 / kSlicedSites;  // per site
            for (int cut = 0; cut < 9; ++cut) {
                const int k0 = cut % 3, k1 = cut / 3;
                cost[cut] = -1;
                if (left_max - k0 - k1 - 2 > 2) continue;
                double bytes = 0, visits = 0;
                for (uint32_t c = 0; c < plan.n_pam; ++c) {
                    double e2 = 0;  // entries of class c's lists of segment 2
                    for (int left = 0; left <= (int)m; ++left) e2 += (double)left_reads[c][left] * nbr(left - k0 - k1 - 2);
                    const double per_list = e2 / kBucketsPerSeg;
                    bytes += (1.0 - std::exp(-per_list)) * (double)genome->ix_class_sites[c] * block_bytes;
                    visits += per_list * (double)genome->ix_class_chunks[c] / kSegments;
                }
                for (int k : {k0, k1}) {  // segments 0 and 1: one list per bucket for all classes
                    const double per_list = (double)reads_any * nbr(k) / kBucketsPerSeg;
                    bytes += (1.0 - std::exp(-per_list)) * sites_all * block_bytes;
                    visits += per_list * chunks_all;
                }
                cost[cut] = std::max(bytes / 4.2e12, visits * 215.0 / ((double)ctx->n_cus * 4 * 2.0e9));
            }
            int best = -1;
            for (int cut = 0; cut < 9; ++cut)
                if (cost[cut] >= 0 && (best < 0 || cost[cut] < cost[best])) best = cut;
            const int forced = ctx->dbg.seed_tight >= 1 ? ctx->dbg.seed_tight - 1 : -1;  // hook: 1 + k0 + 3 k1
            if (forced >= 0 && forced < 9 && cost[forced] >= 0) best = forced;
            if (best < 0) best = 4 * (int)(m ? (m - 1) / kSegments : 0u);  // (no read can reach any class: nothing to search)
            plan.k0 = (uint32_t)(best % 3);
            plan.k1 = (uint32_t)(best / 3);
            k2_max = std::min<int>(2, (int)m - (int)plan.k0 - (int)plan.k1 - 2);
        }
        plan.n_nbr = nbr(std::max<int>(std::max<int>((int)plan.k0, (int)plan.k1), k2_max));
        // entries: one per (read, neighbour within the threshold) of segments 0 and 1, one per class and neighbour of segment 2
        const uint64_t n_pairs = (uint64_t)n_guides * (nbr((int)plan.k0) + nbr((int)plan.k1) + plan.n_pam * nbr(k2_max));
        const uint64_t list_cap = n_pairs + (uint64_t)kLists * (kGuideUnroll - 1) + 2 * kGuideUnroll;
        VSC_HIP_H(ctx->seed_off.ensure((kLists + 1) * sizeof(uint32_t)));
        VSC_HIP_H(ctx->seed_poff.ensure((kLists + 1) * sizeof(uint32_t)));
        VSC_HIP_H(ctx->seed_lrest.ensure(list_cap * sizeof(uint2)));
        VSC_HIP_H(hipMemsetAsync(ctx->seed_lrest.p, 0xFF, list_cap * sizeof(uint2), ctx->stream));  // padding: y = ~0, skipped
        VSC_HIP_H(launch_seed_lists((const uint2 *)ctx->guides.p, n_guides, plan, (uint32_t *)ctx->seed_off.p, (uint32_t *)ctx->seed_poff.p,
                                    (uint2 *)ctx->seed_lrest.p, ctx->stream));
        VSC_HIP_H(hipEventRecord(ctx->ev[7], ctx->stream));
        sa.chunk_tab = genome->d_ix_chunk_tab;
        sa.n_chunks = genome->ix_chunks;
        sa.vert = genome->d_ix_vert;
        sa.list_rest = (const uint2 *)ctx->seed_lrest.p;
        sa.sites = genome->d_ix_sites;
        sa.edge_bits = genome->d_ix_edge;
        sa.guides = (const uint2 *)ctx->guides.p;
        sa.poff = (const uint32_t *)ctx->seed_poff.p;
        sa.max_mm = params->max_mismatches;
        sa.k_half = params->max_mismatches / 2;
        sa.k_seg0 = plan.k0;
        sa.k_seg1 = plan.k1;
        sa.contig_end = genome->d_contig_end;
        sa.n_contigs = genome->n_contigs;
        sa.counters = (unsigned long long *)ctx->counters.p;
        uint32_t groups_per_cu = kSlicedWavesPerSimd;  // resident groups (of four waves) per CU: registers / LDS of the kernel
        if (ctx->dbg.seed_groups_per_cu) groups_per_cu = ctx->dbg.seed_groups_per_cu;
        // dense searches (c3: 129 reads per bucket) share a chunk between the four waves of a workgroup, sparse ones
        // (c2: 13) keep a chunk per wave - see seed_sliced_kernel.  Measured at <= 8 mismatches (tools/
        // experiments.sh shared-threshold): 51 reads per bucket 12.1 vs 11.4 ms, 77: 15.9 vs 16.0, 103: 19.9 vs 20.7
        // (reads per list of segments 0 and 1 - and of segment 2 wherever it is searched as widely)
        seed_shared = (uint64_t)n_guides * nbr((int)std::max(plan.k0, plan.k1)) / kBucketsPerSeg >= 72;
        if (ctx->dbg.seed_shared >= 0) seed_shared = ctx->dbg.seed_shared == 1;
        const uint32_t n_grabs = (sa.n_chunks + kSlicedGrab - 1) / kSlicedGrab;
        const uint32_t n_waves_max = (uint32_t)ctx->n_cus * groups_per_cu * kWavesPerGroup;
        const uint32_t n_waves = std::max<uint32_t>(1, std::min<uint32_t>(n_waves_max, seed_shared ? n_grabs * kWavesPerGroup : n_grabs));
        n_groups = (int)((n_waves + kWavesPerGroup - 1) / kWavesPerGroup);
        // block of records a wave reserves per atomic and region (a power of two, 64 .. 1024): large when many hits
        // are expected, small otherwise (the unused tail of every wave's last block is written as sentinels
        // and read by the sort)
        // with chunk sharing the four waves of a workgroup also share their open output blocks: a quarter of the open
        // lines and of the padding, so the blocks can be eight times as large (c3: per-wave blocks of 128 records 40.6 ms
        // per step, group blocks of 512 39.3, of 1 024 39.0 - tools/experiments.sh group-out; hook seed_group_out = 0 switches it off)
        sa.group_out = seed_shared && ctx->dbg.seed_group_out != 0 ? 1u : 0u;
        const uint32_t owners = sa.group_out ? (uint32_t)n_groups : (uint32_t)n_groups * kWavesPerGroup;  // open blocks per region
        const uint64_t per_wave = cap / ((uint64_t)owners * 8 * n_parts);
        uint32_t want_reserve = (uint32_t)std::min<uint64_t>(n_parts > 8 ? (sa.group_out ? 1024 : 128) : 1024, std::max<uint64_t>(kWave, per_wave));
        if (ctx->dbg.seed_reserve) want_reserve = std::min<uint32_t>(1024, std::max<uint32_t>(kWave, ctx->dbg.seed_reserve));
        sa.reserve_log2 = 6;
        while ((2u << sa.reserve_log2) <= want_reserve) ++sa.reserve_log2;
        sa.reserve = 1u << sa.reserve_log2;
        sa.n_parts = (uint32_t)n_parts;
        sa.pos_pad = pos_pad;
        sa.pos_base = pos_base;
        // a region gets its share of the expected hits + 15 % (read ranges differ) + the open blocks
        part_cap = cap / n_parts + cap / n_parts / 7 + 4096;
        part_cap = std::max<uint64_t>(part_cap, (uint64_t)(1.2 * seen_rate * std::min<uint32_t>(n_guides, kRegionReads)) + 4096);
        part_cap += (uint64_t)owners * sa.reserve;
        cap = part_cap * n_parts;
    }

    ht.lap("prep enqueue");
    for (unsigned tries = 0;; ++tries) {
        // (the kernel keeps a block number in 20 bits, all ones meaning "full": part_cap / reserve < 2^20 - 1)
        if (algo == VSC_ALGO_SEED && (part_cap >= (1ull << 32) - (1u << 20) || part_cap >= ((uint64_t)sa.reserve << 20) - sa.reserve))
            return fail(ctx, VSC_ERR_RANGE, "vsc_search: more than 2^32 hits in a block of 64 reads");
        VSC_HIP_H(ctx->keys_a.ensure(cap * sizeof(uint64_t)));
        VSC_HIP_H(hipMemsetAsync(ctx->counters.p, 0, kCounterWords * sizeof(unsigned long long), ctx->stream));
        VSC_HIP_H(hipEventRecord(ctx->ev[1], ctx->stream));
        if (algo == VSC_ALGO_SCAN) {
            VSC_HIP_H(ctx->vals_a.ensure(cap * sizeof(uint32_t)));
            a.hit_keys = (uint64_t *)ctx->keys_a.p;
            a.hit_vals = (uint32_t *)ctx->vals_a.p;
            a.hit_cap = cap;
            VSC_HIP_H(launch_scan(a, n_groups, false, ctx->stream));
        } else {
            sa.hit_recs = (uint64_t *)ctx->keys_a.p;
            sa.hit_side = nullptr;
            if (want_rows) {
                VSC_HIP_H(ctx->vals_a.ensure(cap * sizeof(uint32_t)));
                sa.hit_side = (uint32_t *)ctx->vals_a.p;
            }
            sa.part_cap = part_cap;
            VSC_HIP_H(launch_seed_sliced(sa, n_groups, seed_shared, ctx->stream));
        }
        VSC_HIP_H(hipEventRecord(ctx->ev[2], ctx->stream));
        VSC_HIP_H(hipMemcpyAsync(cnt, ctx->counters.p, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
        uint32_t list_total = 0;
        if (algo == VSC_ALGO_SEED)
            VSC_HIP_H(hipMemcpyAsync(&list_total, (const uint32_t *)ctx->seed_poff.p + kLists, sizeof list_total, hipMemcpyDeviceToHost, ctx->stream));
        VSC_HIP_H(hipStreamSynchronize(ctx->stream));
        ht.lap("search kernel + sync");
        t.passes++;
        if (algo == VSC_ALGO_SEED) {
            t.list_entries = list_total;
            t.seed_cut = sa.k_seg0 | (sa.k_seg1 << 4);
        }
        if (!cnt[kCntOverflow]) break;
        if (tries >= 2) return fail(ctx, VSC_ERR_DEVICE, "vsc_search: hit buffer overflowed repeatedly");
        // the counters hold the true totals (SEED: records placed + records lost per region, + one block per wave)
        if (algo == VSC_ALGO_SEED) {
            uint64_t need = 0;
            for (int q = 0; q < n_parts; ++q) need = std::max<uint64_t>(need, cnt[kCntPart + 4 * q] + cnt[kCntPart + 4 * q + 2]);
            part_cap = need + (need >> 6) + 4096 + (uint64_t)(sa.group_out ? n_groups : n_groups * kWavesPerGroup) * sa.reserve;
            cap = part_cap * n_parts;
        } else {
            cap = cnt[kCntHits] + (cnt[kCntHits] >> 6) + 4096;
        }
    }
    float ms = 0;
    VSC_HIP_H(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[7]));
    t.prep_ms += ms;
    VSC_HIP_H(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
    t.scan_ms += ms;

    // ---- the segments of the sort: one per region ----------------------------------------------------
    std::vector<SortSeg> segs;
    uint64_t n = 0;
    uint64_t *src = (uint64_t *)ctx->keys_a.p, *other = nullptr;
    SortInfo info;
    if (algo == VSC_ALGO_SCAN) {
        const_cast<vsc_genome *>(genome)->sites = cnt[kCntSites];
        t.sites = cnt[kCntSites];
        t.pairs += cnt[kCntSites] * n_guides;
        n = cnt[kCntHits];
        const_cast<vsc_genome *>(genome)->seen_rate[params->max_mismatches] = (double)n / n_guides;
        if (n >= (1ull << 32)) return fail(ctx, VSC_ERR_RANGE, "vsc_search: more than 2^32 hits in one scan pass (split the read set)");
        if (n > 0) {
            // level 0: (key, value) pairs -> packed records, partitioned by region
            VSC_HIP_H(ctx->keys_b.ensure(cap * sizeof(uint64_t)));
            const unsigned bits0 = ceil_log2((uint64_t)n_parts);
            const size_t n_bins = (size_t)1 << bits0;
            VSC_HIP_H(ctx->sort_segs.ensure(512));
            VSC_HIP_H(ctx->sort_tabs.ensure(3 * n_bins * sizeof(uint32_t)));
            SortSeg s0{0, 0, 0, (uint32_t)n, guide_base};
            const uint32_t tile0[2] = {0u, (uint32_t)((n + kSortTile - 1) / kSortTile)};
            VSC_HIP_H(hipMemcpyAsync(ctx->sort_segs.p, &s0, sizeof s0, hipMemcpyHostToDevice, ctx->stream));
            VSC_HIP_H(hipMemcpyAsync((char *)ctx->sort_segs.p + 256, tile0, sizeof tile0, hipMemcpyHostToDevice, ctx->stream));
            SortArgs l0{};
            l0.segs = (const SortSeg *)ctx->sort_segs.p;
            l0.seg_tile0 = (const uint32_t *)((char *)ctx->sort_segs.p + 256);
            l0.n_segs = 1;
            l0.n_tiles = tile0[1];
            l0.pair_keys = (const uint64_t *)ctx->keys_a.p;
            l0.pair_vals = (const uint32_t *)ctx->vals_a.p;
            l0.out = (uint64_t *)ctx->keys_b.p;
            l0.hist = (uint32_t *)ctx->sort_tabs.p;
            l0.cursor = l0.hist + n_bins;
            l0.bin_start = l0.cursor + n_bins;
            l0.bin_bits = bits0;
            l0.bin_shift = kRecKeyBits;  // key >> 39 = read index >> 6 = region
            l0.pos_pad = pos_pad;
            l0.pos_base = pos_base;
            VSC_HIP_H(hipMemsetAsync(l0.hist, 0, n_bins * sizeof(uint32_t), ctx->stream));
            VSC_HIP_H(launch_bin_hist(l0, ctx->stream));
            VSC_HIP_H(launch_bin_scan(l0, ctx->stream));
            VSC_HIP_H(launch_bin_partition(l0, ctx->stream));
            std::vector<uint32_t> tabs(3 * n_bins);
            VSC_HIP_H(hipMemcpyAsync(tabs.data(), l0.hist, tabs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            VSC_HIP_H(hipStreamSynchronize(ctx->stream));
            for (int q = 0; q < n_parts; ++q) {
                const uint32_t count = tabs[q], start = tabs[2 * n_bins + q];
                if (count) segs.push_back(SortSeg{start, start, used + start, count, guide_base + (uint32_t)q * kRegionReads});
            }
            src = (uint64_t *)ctx->keys_b.p;
            other = (uint64_t *)ctx->keys_a.p;
            info.bytes += 12 * n + 20 * n;  // histogram read; partition read 12, write 8
        }
    } else {
        t.sites = genome->index_sites;
        t.pairs += cnt[kCntSites];
        t.genome_bytes += cnt[kCntVisited] * kVertWords * sizeof(uint32_t) / kSlicedSites;  // sites visited, 3.5 bytes each bit-sliced
        double fullest = 0;
        for (int q = 0; q < n_parts; ++q) {
            const uint64_t placed = cnt[kCntPart + 4 * q], real = placed - cnt[kCntPart + 4 * q + 1];
            fullest = std::max(fullest, (double)placed / std::min<uint32_t>(kRegionReads, n_guides - (uint32_t)q * kRegionReads));
            if (placed) segs.push_back(SortSeg{(uint64_t)q * part_cap, (uint64_t)q * part_cap, used + n, (uint32_t)placed,
                                               guide_base + (uint32_t)q * kRegionReads});
            n += real;
        }
        const_cast<vsc_genome *>(genome)->seen_rate[params->max_mismatches] = fullest;
        // (the sort's second buffer, ctx->keys_b, is sized by bin_sort for the layout its first level uses)
    }
    t.hits += n;
    if (n > 0) {
        ht.lap("sort buffers");
        VSC_HIP_H(result_room(ctx, hits, used, n, projected));
        ht.lap("record storage");
        const unsigned key_bits = pos_bits + 1 + ceil_log2(std::min<uint32_t>(n_guides, kRegionReads));
        // the seed search's regions go through the slot partition (no histogram pass) unless this genome has shown
        // bins that outgrow their slots at this budget
        bool *slots = algo == VSC_ALGO_SEED ? &const_cast<vsc_genome *>(genome)->sort_slots_ok[params->max_mismatches] : nullptr;
        SortRows rows;
        if (want_rows) {
            // the batch's rows, all at once (the caller reads them in its callback): 64 bytes per hit - 104 GB for 10 000 reads at
            // 8 mismatches on 3 Gbp.  If that does not fit beside the pooled buffers, those go back first.
            if (ctx->score_feat.ensure(n * VSC_PACKED_FEATURE_BYTES) != hipSuccess) {
                (void)hipGetLastError();
                for (auto &b : ctx->spare_records) b.release();
                ctx->spare_records.clear();
                for (DeviceBuf *b : {&ctx->score_mit, &ctx->score_flags, &ctx->score_sched, &ctx->keys_b, &ctx->vals_b}) b->release();
                VSC_HIP_H(ctx->score_feat.ensure(n * VSC_PACKED_FEATURE_BYTES));
            }
            rows.side_src = (uint32_t *)ctx->vals_a.p;
            rows.side_other = &ctx->vals_b;
            rows.guides = (const uint2 *)ctx->guides.p;
            rows.guide_first = guide_base;
            rows.rows = (uint4 *)ctx->score_feat.p;
            rows.rows_first = used;
        }
        VSC_HIP_H(bin_sort(ctx, genome, std::move(segs), src, other, key_bits, pos_pad, pos_base, hits->d_records, ctx->ev[3], &info,
                           algo == VSC_ALGO_SEED ? &ctx->keys_b : nullptr, slots, want_rows ? &rows : nullptr));
    } else {
        VSC_HIP_H(hipEventRecord(ctx->ev[3], ctx->stream));
    }
    VSC_HIP_H(hipEventRecord(ctx->ev[4], ctx->stream));
    VSC_HIP_H(hipStreamSynchronize(ctx->stream));
    ht.lap("sort + finalize + sync");
    VSC_HIP_H(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
    t.sort_ms += ms;
    VSC_HIP_H(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
    t.finalize_ms += ms;
    VSC_HIP_H(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[4]));
    t.total_ms += ms;
    t.sort_levels = std::max(t.sort_levels, info.levels);
    t.sort_bin_bits = std::max(t.sort_bin_bits, info.bin_bits);
    t.sort_bytes += info.bytes;
    t.sort_fallbacks += info.slot_fallbacks;
    t.read_passes++;
    res->n = n;
    return VSC_OK;
#undef VSC_HIP_H
}

// common front end of vsc_search / vsc_search_stream: argument checks, choice of the algorithm, index
int search_setup(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
                 const vsc_search_params *params, const char *who, int *algo_out, vsc_timing *t)
{
    ctx->err.clear();
    if (!genome || !params || (n_guides && !guides)) return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": null argument").c_str());
    if (genome->ctx != ctx) return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": genome belongs to another context").c_str());
    if (params->max_mismatches > VSC_MAX_MISMATCHES)  // read_mapping/bidir_mapping.cpp:234-238
        return fail(ctx, VSC_ERR_INVALID, "Maximum number of mismatches must lie between 0 and 8.");
    if (params->algorithm > VSC_ALGO_SEED) return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": unknown algorithm").c_str());
    if (n_guides >= (1u << 31)) return fail(ctx, VSC_ERR_RANGE, (std::string(who) + ": too many reads").c_str());
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    *t = vsc_timing{};
    t->index_ms = ctx->timing.index_ms;
    const double own_bases = (double)genome->n_tiles * kTileBases;
    int algo = params->algorithm;
    if (algo == VSC_ALGO_AUTO) {
        // the index costs about as much as scanning a few hundred reads; it pays for itself when it is
        // already there or when the search is big enough
        const bool worth_building = (double)n_guides * own_bases >= 2.0e10;
        algo = (index_matches(genome, params) || worth_building) ? VSC_ALGO_SEED : VSC_ALGO_SCAN;
    }
    if (n_guides && algo == VSC_ALGO_SEED && !index_matches(genome, params)) {
        std::string why;
        hipError_t e = build_index(ctx, const_cast<vsc_genome *>(genome), params, &why);
        if (e != hipSuccess) {
            if (params->algorithm == VSC_ALGO_SEED || why.empty())
                return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE,
                            why.empty() ? "vsc_search: building the seed index" : why.c_str(), why.empty() ? e : hipSuccess);
            algo = VSC_ALGO_SCAN;  // auto mode: a genome too large for the index is still searchable
        } else {
            t->index_ms = genome->index_ms;
        }
    }
    t->algorithm = (uint32_t)algo;
    *algo_out = algo;
    return VSC_OK;
}

}  // namespace

extern "C" {

int vsc_search(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
               const vsc_search_params *params, vsc_hits **out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    vsc_timing t{};
    int algo = 0;
    const int rc = search_setup(ctx, genome, guides, n_guides, params, "vsc_search", &algo, &t);
    if (rc != VSC_OK) return rc;
    vsc_hits *hits = new (std::nothrow) vsc_hits();
    if (!hits) return fail(ctx, VSC_ERR_NOMEM, "vsc_search: out of host memory");
    hits->ctx = ctx;
    // a pass takes at most kMaxPassReads reads (128 output regions of 64 reads); larger sets are searched
    // pass by pass - the read index is the major sort key, so the passes' results simply follow each other
    uint64_t used = 0;
    for (uint32_t first = 0; first < n_guides; first += kMaxPassReads) {
        const uint32_t count = std::min<uint32_t>(kMaxPassReads, n_guides - first);
        const uint64_t projected = first ? (uint64_t)((double)used / first * n_guides * 1.02) + 4096 : 0;
        PassResult r;
        const int prc = search_pass(ctx, genome, guides + first, count, first, params, algo, hits, used, projected, t, &r);
        if (prc != VSC_OK) {
            vsc_hits_free(hits);
            return prc;
        }
        used += r.n;
    }
    hits->n = used;
    if (used == 0) hits->host_valid = true;
    ctx->timing = t;
    *out = hits;
    return VSC_OK;
    });
}

int vsc_search_stream(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
                      const vsc_search_params *params, uint32_t batch_reads, vsc_batch_fn on_batch, void *user)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    vsc_timing t{};
    int algo = 0;
    const int rc = search_setup(ctx, genome, guides, n_guides, params, "vsc_search_stream", &algo, &t);
    if (rc != VSC_OK) return rc;
    if (!on_batch) return fail(ctx, VSC_ERR_INVALID, "vsc_search_stream: null callback");
    if (batch_reads == 0 || batch_reads > (uint32_t)kMaxPassReads) batch_reads = kMaxPassReads;
    for (uint32_t first = 0; first < n_guides; first += batch_reads) {
        const uint32_t count = std::min<uint32_t>(batch_reads, n_guides - first);
        vsc_hits *hits = new (std::nothrow) vsc_hits();
        if (!hits) return fail(ctx, VSC_ERR_NOMEM, "vsc_search_stream: out of host memory");
        hits->ctx = ctx;
        PassResult r;
        int prc = search_pass(ctx, genome, guides + first, count, first, params, algo, hits, 0, 0, t, &r);
        if (prc == VSC_OK) {
            hits->n = r.n;
            if (r.n == 0) hits->host_valid = true;
            ctx->timing = t;
            ctx->timing.score_ms = 0;
            prc = on_batch(user, hits, first, count);
            t.score_ms += ctx->timing.score_ms;  // what the callback's scoring calls measured
            if (prc != VSC_OK && ctx->err.empty()) ctx->err = "vsc_search_stream: the batch callback failed";
        }
        vsc_hits_free(hits);
        if (prc != VSC_OK) return prc;
    }
    ctx->timing = t;
    return VSC_OK;
    });
}

int vsc_search_stream_rows(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
                           const vsc_search_params *params, uint32_t batch_reads, vsc_rows_batch_fn on_batch, void *user)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    vsc_timing t{};
    int algo = 0;
    const int rc = search_setup(ctx, genome, guides, n_guides, params, "vsc_search_stream_rows", &algo, &t);
    if (rc != VSC_OK) return rc;
    if (!on_batch) return fail(ctx, VSC_ERR_INVALID, "vsc_search_stream_rows: null callback");
    if (batch_reads == 0 || batch_reads > (uint32_t)kMaxPassReads) batch_reads = kMaxPassReads;
    for (uint32_t first = 0; first < n_guides; first += batch_reads) {
        const uint32_t count = std::min<uint32_t>(batch_reads, n_guides - first);
        vsc_hits *hits = new (std::nothrow) vsc_hits();
        if (!hits) return fail(ctx, VSC_ERR_NOMEM, "vsc_search_stream_rows: out of host memory");
        hits->ctx = ctx;
        PassResult r;
        // the seed search hands the sites' bases to the record assembly, which writes the rows; the streaming scan's hits
        // (small searches on a genome without an index) are scored the usual way afterwards
        int prc = search_pass(ctx, genome, guides + first, count, first, params, algo, hits, 0, 0, t, &r, algo == VSC_ALGO_SEED);
        if (prc == VSC_OK) {
            hits->n = r.n;
            if (r.n == 0) hits->host_valid = true;
            ctx->timing = t;
            ctx->timing.score_ms = 0;
            const void *rows = nullptr;
            if (r.n && algo != VSC_ALGO_SEED) {
                prc = vsc_score_hits_packed(ctx, genome, hits, guides, n_guides, 0, r.n, nullptr, nullptr, nullptr);
                if (prc == VSC_OK && ctx->score_feat.cap < r.n * VSC_PACKED_FEATURE_BYTES)
                    prc = fail(ctx, VSC_ERR_NOMEM, "vsc_search_stream_rows: the batch's rows do not fit the device at once (smaller batches)");
                t.score_ms += ctx->timing.score_ms;
            }
            if (prc == VSC_OK) {
                if (r.n) rows = ctx->score_feat.p;
                prc = on_batch(user, hits, first, count, rows);
                if (prc != VSC_OK && ctx->err.empty()) ctx->err = "vsc_search_stream_rows: the batch callback failed";
            }
        }
        vsc_hits_free(hits);
        if (prc != VSC_OK) return prc;
    }
    ctx->timing = t;
    return VSC_OK;
    });
}

uint64_t vsc_hits_count(const vsc_hits *hits) { return hits ? hits->n : 0; }

const void *vsc_hits_data_dev(const vsc_hits *hits) { return hits ? hits->d_records : nullptr; }

int vsc_hits_data(vsc_hits *hits, const vsc_hit **out)
{
    return guarded(nullptr, [&]() -> int {
    if (!hits || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    if (!hits->host_valid) {
        vsc_ctx *ctx = hits->ctx;
        try {
            hits->host.resize(hits->n);
        } catch (...) {
            return fail(ctx, VSC_ERR_NOMEM, "vsc_hits_data: out of host memory");
        }
        if (hits->n) {
            VSC_HIP(ctx, hipSetDevice(ctx->device));
            VSC_HIP(ctx, hipMemcpyAsync(hits->host.data(), hits->d_records, hits->n * sizeof(vsc_hit),
                                        hipMemcpyDeviceToHost, ctx->stream));
            VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        hits->host_valid = true;
    }
    *out = hits->host.data();
    return VSC_OK;
    });
}

int vsc_hits_copy(vsc_hits *hits, void *dst, int dst_is_device)
{
    return guarded(nullptr, [&]() -> int {
    if (!hits || (!dst && hits->n)) return VSC_ERR_INVALID;
    if (hits->n == 0) return VSC_OK;
    vsc_ctx *ctx = hits->ctx;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    VSC_HIP(ctx, hipMemcpyAsync(dst, hits->d_records, hits->n * sizeof(vsc_hit),
                                dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VSC_OK;
    });
}

int vsc_hits_merge(vsc_ctx *ctx, const void *records, int records_on_device, const uint64_t *shard_counts,
                   uint32_t n_shards, uint32_t n_guides, vsc_hits **out)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    ctx->err.clear();
    if (!shard_counts || n_shards == 0) return fail(ctx, VSC_ERR_INVALID, "vsc_hits_merge: null argument");
    std::vector<uint64_t> off(n_shards + 1, 0);
    for (uint32_t s = 0; s < n_shards; ++s) off[s + 1] = off[s] + shard_counts[s];
    const uint64_t n = off[n_shards];
    if (n && !records) return fail(ctx, VSC_ERR_INVALID, "vsc_hits_merge: null argument");
    if (n_guides >= (1u << 30)) return fail(ctx, VSC_ERR_RANGE, "vsc_hits_merge: too many reads");
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    vsc_hits *hits = new (std::nothrow) vsc_hits();
    if (!hits) return fail(ctx, VSC_ERR_NOMEM, "vsc_hits_merge: out of host memory");
    hits->ctx = ctx;
    hits->n = n;
    if (n == 0) {
        hits->host_valid = true;
        *out = hits;
        return VSC_OK;
    }
    const uint32_t K = 2 * std::max<uint32_t>(n_guides, 1);  // keys guide << 1 | strand
    const uint64_t nb = (uint64_t)n_shards * (K + 1);
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    // small work arrays reuse the search scratch buffers
    step(ctx->keys_b.ensure((nb + n_shards + 1) * sizeof(uint64_t)));
    step(ctx->vals_a.ensure(((uint64_t)K + 1) * sizeof(uint64_t)));
    step(take_records(ctx, hits, n));
    const vsc_hit *records_dev = (const vsc_hit *)records;
    if (!records_on_device) {
        step(ctx->score_feat.ensure(n * sizeof(vsc_hit)));  // staging buffer for host input
        if (e == hipSuccess)
            step(hipMemcpyAsync(ctx->score_feat.p, records, n * sizeof(vsc_hit), hipMemcpyHostToDevice, ctx->stream));
        records_dev = (const vsc_hit *)ctx->score_feat.p;
    }
    uint64_t *bound = (uint64_t *)ctx->keys_b.p;
    uint64_t *shard_off_dev = bound + nb;
    if (e == hipSuccess)
        step(hipMemcpyAsync(shard_off_dev, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    step(hipEventRecord(ctx->ev[0], ctx->stream));
    if (e == hipSuccess)
        step(launch_merge(records_dev, shard_off_dev, n_shards, K, bound, (uint64_t *)ctx->vals_a.p, hits->d_records, ctx->stream));
    step(hipEventRecord(ctx->ev[1], ctx->stream));
    step(hipStreamSynchronize(ctx->stream));
    if (e != hipSuccess) {
        vsc_hits_free(hits);
        return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, "vsc_hits_merge", e);
    }
    *out = hits;
    return VSC_OK;
    });
}

int vsc_hits_pack_exchange(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, uint32_t n_guides, void *records,
                           int records_on_device, uint32_t *key_counts)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (!genome || !hits || genome->ctx != ctx || hits->ctx != ctx || (n_guides && !key_counts) || (hits->n && !records))
        return fail(ctx, VSC_ERR_INVALID, "vsc_hits_pack_exchange: null argument or object of another context");
    if (n_guides >= (1u << 30)) return fail(ctx, VSC_ERR_RANGE, "vsc_hits_pack_exchange: too many reads");
    const uint32_t K = 2 * n_guides;
    const uint64_t n = hits->n;
    if (n == 0) {
        std::fill(key_counts, key_counts + K, 0u);
        return VSC_OK;
    }
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    VSC_HIP(ctx, ctx->sort_tabs.ensure(((size_t)K + 1 + 2) * sizeof(uint64_t)));
    uint64_t *d_bound = (uint64_t *)ctx->sort_tabs.p, *d_range = d_bound + K + 1;
    const uint64_t range[2] = {0, n};
    VSC_HIP(ctx, hipMemcpyAsync(d_range, range, sizeof range, hipMemcpyHostToDevice, ctx->stream));
    VSC_HIP(ctx, launch_key_bounds(hits->d_records, d_range, K, d_bound, ctx->stream));
    uint64_t *dst = (uint64_t *)records;
    if (!records_on_device) {
        VSC_HIP(ctx, ctx->score_feat.ensure(n * sizeof(uint64_t)));
        dst = (uint64_t *)ctx->score_feat.p;
    }
    VSC_HIP(ctx, launch_xpack(hits->d_records, n, genome->d_contig_off, dst, ctx->stream));
    std::vector<uint64_t> bound((size_t)K + 1);
    VSC_HIP(ctx, hipMemcpyAsync(bound.data(), d_bound, bound.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (!records_on_device) VSC_HIP(ctx, hipMemcpyAsync(records, dst, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bound[0] != 0 || bound[K] != n) return fail(ctx, VSC_ERR_INVALID, "vsc_hits_pack_exchange: the result holds reads beyond n_guides");
    for (uint32_t k = 0; k < K; ++k) {
        const uint64_t c = bound[k + 1] - bound[k];
        if (c >= (1ull << 32)) return fail(ctx, VSC_ERR_RANGE, "vsc_hits_pack_exchange: more than 2^32 hits of one read and strand");
        key_counts[k] = (uint32_t)c;
    }
    return VSC_OK;
    });
}

}  // extern "C"

namespace {
// the common body of vsc_hits_merge_packed / _votes: records (and votes) of all shards concatenated, host or device
int merge_packed_concat(vsc_ctx *ctx, const vsc_genome *genome, const void *records, const void *votes, int on_device, const uint32_t *key_counts,
                        uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out, void *votes_out, int votes_out_on_device,
                        const char *who)
{
    if (!ctx || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    ctx->err.clear();
    const std::string w(who);
    if (!genome || genome->ctx != ctx || n_shards == 0 || (n_keys && !key_counts))
        return fail(ctx, VSC_ERR_INVALID, (w + ": null argument or genome of another context").c_str());
    if ((uint64_t)first_key + n_keys > (1ull << 31) || (uint64_t)n_keys * n_shards >= (1ull << 31))
        return fail(ctx, VSC_ERR_RANGE, (w + ": too many keys").c_str());
    uint64_t n = 0;
    std::vector<uint64_t> shard_n(n_shards, 0);
    for (uint32_t s = 0; s < n_shards; ++s) {
        for (uint32_t k = 0; k < n_keys; ++k) shard_n[s] += key_counts[(size_t)s * n_keys + k];
        n += shard_n[s];
    }
    if (n && (!records || (votes_out && !votes))) return fail(ctx, VSC_ERR_INVALID, (w + ": null records").c_str());
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t *records_dev = (const uint64_t *)records;
    const uint16_t *votes_dev = (const uint16_t *)votes;
    if (n && !on_device) {  // staging buffers for host input
        VSC_HIP(ctx, ctx->score_feat.ensure(n * sizeof(uint64_t)));
        VSC_HIP(ctx, hipMemcpyAsync(ctx->score_feat.p, records, n * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        records_dev = (const uint64_t *)ctx->score_feat.p;
        if (votes) {
            VSC_HIP(ctx, ctx->score_flags.ensure(n * sizeof(uint16_t)));
            VSC_HIP(ctx, hipMemcpyAsync(ctx->score_flags.p, votes, n * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
            votes_dev = (const uint16_t *)ctx->score_flags.p;
        }
    }
    std::vector<const void *> shard_ptr(n_shards), side_ptr(n_shards);
    uint64_t at = 0;
    for (uint32_t s = 0; s < n_shards; ++s) {
        shard_ptr[s] = records_dev + at;
        side_ptr[s] = votes_dev ? votes_dev + at : nullptr;
        at += shard_n[s];
    }
    DeviceBuf side;  // the merged votes, before they go where the caller wants them
    const int rc = vsc::merge_packed_shards(ctx, genome, shard_ptr.data(), votes ? side_ptr.data() : nullptr, key_counts, n_shards, first_key,
                                            n_keys, out, votes && votes_out ? &side : nullptr);
    if (rc == VSC_OK && n && votes && votes_out) {
        hipError_t e = hipMemcpyAsync(votes_out, side.p, n * sizeof(uint16_t), votes_out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                      ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            side.release();
            vsc_hits_free(*out);
            *out = nullptr;
            return fail(ctx, VSC_ERR_DEVICE, who, e);
        }
    }
    side.release();
    return rc;
}
}  // namespace

extern "C" {

int vsc_hits_merge_packed(vsc_ctx *ctx, const vsc_genome *genome, const void *records, int records_on_device, const uint32_t *key_counts,
                          uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out)
{
    return guarded(ctx, [&]() -> int {
        return merge_packed_concat(ctx, genome, records, nullptr, records_on_device, key_counts, n_shards, first_key, n_keys, out, nullptr, 0,
                                   "vsc_hits_merge_packed");
    });
}

int vsc_hits_merge_packed_votes(vsc_ctx *ctx, const vsc_genome *genome, const void *records, const void *votes, int on_device,
                                const uint32_t *key_counts, uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out,
                                void *votes_out, int votes_out_on_device)
{
    return guarded(ctx, [&]() -> int {
        return merge_packed_concat(ctx, genome, records, votes, on_device, key_counts, n_shards, first_key, n_keys, out, votes_out,
                                   votes_out_on_device, "vsc_hits_merge_packed_votes");
    });
}

}  // extern "C"

// The merge of vsc_hits_merge_packed over per-shard record buffers (device memory of ctx's device, shard s's records for the
// keys [first_key, first_key + n_keys) at shard_records[s]) - what the multi-device search hands over: every shard's records
// arrive in a buffer of their own, whenever that shard is done.  shard_side (optional): 2 bytes per record that travel with
// it (the shard's classifier votes); they arrive in side_out, in the order of the merged records.
int vsc::merge_packed_shards(vsc_ctx *ctx, const vsc_genome *genome, const void *const *shard_records, const void *const *shard_side,
                             const uint32_t *key_counts, uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out,
                             DeviceBuf *side_out)
{
    // where every (key, shard) segment lies, and where it goes
    const size_t n_segs = (size_t)n_keys * n_shards;
    std::vector<uint64_t> seg_src(n_segs), seg_dst(n_segs), seg_side(shard_side ? n_segs : 0);
    std::vector<uint32_t> seg_n(n_segs);
    for (uint32_t s = 0; s < n_shards; ++s) {
        uint64_t at = 0;
        for (uint32_t k = 0; k < n_keys; ++k) {
            const size_t seg = (size_t)k * n_shards + s;
            seg_src[seg] = (uint64_t)(uintptr_t)shard_records[s] + at * sizeof(uint64_t);
            if (shard_side) seg_side[seg] = (uint64_t)(uintptr_t)shard_side[s] + at * sizeof(uint16_t);
            at += key_counts[(size_t)s * n_keys + k];
        }
    }
    uint64_t n = 0;
    for (uint32_t k = 0; k < n_keys; ++k)
        for (uint32_t s = 0; s < n_shards; ++s) {
            const size_t seg = (size_t)k * n_shards + s;
            seg_n[seg] = key_counts[(size_t)s * n_keys + k];
            seg_dst[seg] = n;
            n += seg_n[seg];
        }
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    vsc_hits *hits = new (std::nothrow) vsc_hits();
    if (!hits) return fail(ctx, VSC_ERR_NOMEM, "vsc_hits_merge_packed: out of host memory");
    hits->ctx = ctx;
    hits->n = n;
    if (n == 0) {
        hits->host_valid = true;
        *out = hits;
        return VSC_OK;
    }
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
    };
    const size_t tab_bytes = n_segs * (3 * sizeof(uint64_t) + sizeof(uint32_t)) + 2 * sizeof(uint32_t);
    step(ctx->keys_b.ensure(tab_bytes));
    step(take_records(ctx, hits, n));
    if (side_out) step(side_out->ensure(n * sizeof(uint16_t)));
    uint64_t *d_src = (uint64_t *)ctx->keys_b.p, *d_dst = d_src + n_segs, *d_side = d_dst + n_segs;
    uint32_t *d_n = (uint32_t *)(d_side + n_segs);
    uint32_t *d_bad = d_n + n_segs;  // records whose position lies in no contig or that do not ascend inside a segment
    uint32_t n_bad = 0;
    if (e == hipSuccess) {
        step(hipMemcpyAsync(d_src, seg_src.data(), n_segs * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        step(hipMemcpyAsync(d_dst, seg_dst.data(), n_segs * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        if (shard_side) step(hipMemcpyAsync(d_side, seg_side.data(), n_segs * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        step(hipMemcpyAsync(d_n, seg_n.data(), n_segs * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        step(hipMemsetAsync(d_bad, 0, sizeof(uint32_t), ctx->stream));
        step(launch_merge_packed(d_src, d_dst, d_n, (uint32_t)n_segs, n_shards, first_key, genome->d_contig_off, genome->d_contig_end,
                                 genome->n_contigs, hits->d_records, d_bad, shard_side ? d_side : nullptr,
                                 shard_side && side_out ? (uint16_t *)side_out->p : nullptr, ctx->stream));
        step(hipMemcpyAsync(&n_bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        step(hipStreamSynchronize(ctx->stream));  // (the host tables go out of scope)
    }
    if (e != hipSuccess) {
        vsc_hits_free(hits);
        return fail(ctx, e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, "vsc_hits_merge_packed", e);
    }
    if (n_bad) {  // what a peer or the caller sent is not a sorted shard result of this genome
        vsc_hits_free(hits);
        return fail(ctx, VSC_ERR_INVALID, ("vsc_hits_merge_packed: " + std::to_string(n_bad) +
                    " exchange records lie in no contig of the genome or do not ascend inside their (key, shard) segment").c_str());
    }
    *out = hits;
    return VSC_OK;
}

extern "C" {

int vsc_hits_free(vsc_hits *hits)
{
    if (!hits) return VSC_OK;
    if (hits->storage.p) {
        vsc_ctx *ctx = hits->ctx;
        if (ctx->spare_records.size() < 2) {
            ctx->spare_records.push_back(hits->storage);  // reused by the next search
        } else {
            (void)hipSetDevice(ctx->device);
            hits->storage.release();
        }
    }
    delete hits;
    return VSC_OK;
}

// ------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// Rows of scratch a scoring call gets: as many as fit.  Halves the request until the device has room (down
// to 64 Ki rows), so that a result larger than the free memory is scored in several passes over the same
// scratch buffers.  vsc_debug_params.score_chunk (tests) forces passes of at most that many rows.
hipError_t score_scratch(vsc_ctx *ctx, uint64_t count, size_t mit_bytes, size_t flag_bytes, size_t feat_bytes, uint64_t *rows_out)
{
    uint64_t rows = count;
    if (ctx->dbg.score_chunk) rows = std::min<uint64_t>(rows, ctx->dbg.score_chunk);
    for (;;) {
        hipError_t e = hipSuccess;
        if (mit_bytes) e = ctx->score_mit.ensure(rows * mit_bytes);
        if (e == hipSuccess && flag_bytes) e = ctx->score_flags.ensure(rows * flag_bytes);
        if (e == hipSuccess && feat_bytes) e = ctx->score_feat.ensure(rows * feat_bytes);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || rows <= (1u << 16)) return e;
        // give back what this attempt got before asking for less
        ctx->score_mit.release();
        ctx->score_flags.release();
        ctx->score_feat.release();
        rows = (rows + 1) / 2;
    }
    *rows_out = rows;
    return hipSuccess;
}

// Word-wise hash of a host array (four independent multiply-xorshift lanes over its 8-byte words + the tail bytes): what the
// caches below key on - an order of magnitude cheaper than walking the bytes, and called once per scoring call.
uint64_t hash_words(const void *p, size_t bytes, uint64_t seed)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h[4] = {seed, seed ^ 0x9E3779B97F4A7C15ull, seed + 0xBF58476D1CE4E5B9ull, ~seed};
    const size_t nw = bytes / 8;
    size_t i = 0;
    for (; i + 4 <= nw; i += 4)
        for (int k = 0; k < 4; ++k) {
            uint64_t w;
            std::memcpy(&w, b + 8 * (i + k), 8);
            h[k] = (h[k] ^ w) * 0x100000001b3ull;
            h[k] ^= h[k] >> 29;
        }
    uint64_t tail = 0xcbf29ce484222325ull;
    for (size_t q = 8 * i; q < bytes; ++q) tail = (tail ^ b[q]) * 0x100000001b3ull;
    uint64_t r = tail ^ (uint64_t)bytes;
    for (int k = 0; k < 4; ++k) r = (r ^ h[k]) * 0xff51afd7ed558ccdull, r ^= r >> 33;
    return r;
}

// The reads as (hi, lo) plane pairs for the scoring kernels, in a buffer of their own (ctx->guides is the search passes'
// and changes with every batch of a streamed search): uploaded when the read set differs from the last call's.
hipError_t upload_read_planes(vsc_ctx *ctx, const uint64_t *guides, uint32_t n_guides)
{
    const uint64_t h = hash_words(guides, (size_t)n_guides * sizeof(uint64_t), 0x5c0 + n_guides);
    if (ctx->score_guides.p && ctx->score_guides_n == n_guides && ctx->score_guides_hash == h) return hipSuccess;
    ctx->score_guides_n = ~0u;
    std::vector<uint32_t> gp((size_t)std::max<uint32_t>(n_guides, 1) * 2, 0);
    for (uint32_t i = 0; i < n_guides; ++i) guide_planes(guides[i], &gp[2 * (size_t)i], &gp[2 * (size_t)i + 1]);
    VSC_TRY(ctx->score_guides.ensure(gp.size() * sizeof(uint32_t)));
    VSC_TRY(hipMemcpyAsync(ctx->score_guides.p, gp.data(), gp.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VSC_TRY(hipStreamSynchronize(ctx->stream));  // gp goes out of scope
    ctx->score_guides_n = n_guides;
    ctx->score_guides_hash = h;
    return hipSuccess;
}

void fill_score_args(ScoreArgs &s, vsc_ctx *ctx, const vsc_genome *genome)
{
    s.hl = genome->d_hl;
    s.first_pos = (uint32_t)(genome->first_word * 32);
    s.n_plane_words = genome->dev_words;
    s.contig_off = genome->d_contig_off;
    s.guides = (const uint2 *)ctx->score_guides.p;
}

}  // namespace

extern "C" {

int vsc_score_hits(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, const uint64_t *guides,
                   uint32_t n_guides, uint64_t first, uint64_t count, double *mit, uint8_t *mit_flags,
                   uint8_t *features)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (!genome || !hits || (n_guides && !guides)) return fail(ctx, VSC_ERR_INVALID, "vsc_score_hits: null argument");
    if (first > hits->n || count > hits->n - first) return fail(ctx, VSC_ERR_INVALID, "vsc_score_hits: row range outside the result");
    ctx->timing.score_ms = 0;
    if (count == 0 || (!mit && !mit_flags && !features)) return VSC_OK;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    VSC_HIP(ctx, upload_read_planes(ctx, guides, n_guides));
    VSC_HIP(ctx, ensure_hl(ctx, genome));
    uint64_t rows = 0;
    VSC_HIP(ctx, score_scratch(ctx, count, mit ? sizeof(double) : 0, mit_flags ? 1 : 0, features ? VSC_N_FEATURES : 0, &rows));
    double total_ms = 0;
    for (uint64_t done = 0; done < count; done += rows) {  // one pass unless the scratch buffers had to be smaller than the result
        const uint64_t m = std::min(rows, count - done);
        ScoreArgs s{};
        fill_score_args(s, ctx, genome);
        s.hits = hits->d_records + first + done;
        s.n = m;
        if (mit) s.mit = (double *)ctx->score_mit.p;
        if (mit_flags) s.mit_flags = (uint8_t *)ctx->score_flags.p;
        if (features) s.features = (uint8_t *)ctx->score_feat.p;
        VSC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        VSC_HIP(ctx, launch_score(s, ctx->stream));
        VSC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        if (mit) VSC_HIP(ctx, hipMemcpyAsync(mit + done, s.mit, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (mit_flags) VSC_HIP(ctx, hipMemcpyAsync(mit_flags + done, s.mit_flags, m, hipMemcpyDeviceToHost, ctx->stream));
        if (features)
            VSC_HIP(ctx, hipMemcpyAsync(features + done * VSC_N_FEATURES, s.features, m * VSC_N_FEATURES, hipMemcpyDeviceToHost, ctx->stream));
        VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        VSC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        total_ms += ms;
    }
    ctx->timing.score_ms = total_ms;
    return VSC_OK;
    });
}

int vsc_score_pairs(vsc_ctx *ctx, const uint64_t *on_targets, const uint64_t *off_targets, const uint32_t *masks,
                    uint64_t n, double *mit, uint8_t *mit_flags, uint8_t *features)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (n && (!on_targets || !off_targets || !masks)) return fail(ctx, VSC_ERR_INVALID, "vsc_score_pairs: null argument");
    ctx->timing.score_ms = 0;
    if (n == 0 || (!mit && !mit_flags && !features)) return VSC_OK;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> planes((size_t)n * 4);
    for (uint64_t i = 0; i < n; ++i) {
        guide_planes(on_targets[i], &planes[2 * i], &planes[2 * i + 1]);
        guide_planes(off_targets[i], &planes[2 * (n + i)], &planes[2 * (n + i) + 1]);
    }
    VSC_HIP(ctx, ctx->guides.ensure(planes.size() * sizeof(uint32_t) + n * sizeof(uint32_t)));
    uint32_t *d_planes = (uint32_t *)ctx->guides.p;
    uint32_t *d_masks = d_planes + planes.size();
    VSC_HIP(ctx, hipMemcpyAsync(d_planes, planes.data(), planes.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VSC_HIP(ctx, hipMemcpyAsync(d_masks, masks, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    double *d_mit = nullptr;
    uint8_t *d_flags = nullptr, *d_feat = nullptr;
    if (mit) {
        VSC_HIP(ctx, ctx->score_mit.ensure(n * sizeof(double)));
        d_mit = (double *)ctx->score_mit.p;
    }
    if (mit_flags) {
        VSC_HIP(ctx, ctx->score_flags.ensure(n));
        d_flags = (uint8_t *)ctx->score_flags.p;
    }
    if (features) {
        VSC_HIP(ctx, ctx->score_feat.ensure(n * VSC_N_FEATURES));
        d_feat = (uint8_t *)ctx->score_feat.p;
    }
    VSC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    VSC_HIP(ctx, launch_score_pairs((const uint2 *)d_planes, (const uint2 *)(d_planes + 2 * n), d_masks, n, d_mit, d_flags,
                                    d_feat, ctx->stream));
    VSC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (mit) VSC_HIP(ctx, hipMemcpyAsync(mit, d_mit, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (mit_flags) VSC_HIP(ctx, hipMemcpyAsync(mit_flags, d_flags, n, hipMemcpyDeviceToHost, ctx->stream));
    if (features) VSC_HIP(ctx, hipMemcpyAsync(features, d_feat, n * VSC_N_FEATURES, hipMemcpyDeviceToHost, ctx->stream));
    VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    VSC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timing.score_ms = ms;
    return VSC_OK;
    });
}

int vsc_score_hits_packed(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, const uint64_t *guides,
                          uint32_t n_guides, uint64_t first, uint64_t count, void *packed_dev, uint32_t *packed_host,
                          double *mit_host)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (!genome || !hits || (n_guides && !guides)) return fail(ctx, VSC_ERR_INVALID, "vsc_score_hits_packed: null argument");
    if (first > hits->n || count > hits->n - first) return fail(ctx, VSC_ERR_INVALID, "vsc_score_hits_packed: row range outside the result");
    ctx->timing.score_ms = 0;
    if (count == 0) return VSC_OK;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    VSC_HIP(ctx, upload_read_planes(ctx, guides, n_guides));
    VSC_HIP(ctx, ensure_hl(ctx, genome));
    // rows go to the caller's device buffer (whole result, one pass unless the MIT scratch is short) or through
    // library scratch, as many rows per pass as fit (c3's 1.6e9 rows are 104 GB: several passes)
    uint64_t rows = 0;
    VSC_HIP(ctx, score_scratch(ctx, count, mit_host ? sizeof(double) : 0, 0, packed_dev ? 0 : VSC_PACKED_FEATURE_BYTES, &rows));
    double total_ms = 0;
    for (uint64_t done = 0; done < count; done += rows) {
        const uint64_t m = std::min(rows, count - done);
        ScoreArgs s{};
        fill_score_args(s, ctx, genome);
        s.hits = hits->d_records + first + done;
        s.n = m;
        if (mit_host) s.mit = (double *)ctx->score_mit.p;
        uint4 *dst = packed_dev ? (uint4 *)packed_dev + done * 4 : (uint4 *)ctx->score_feat.p;
        VSC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        // large results on a large genome: visit the rows genome slice by genome slice (launch_score_schedule)
        uint32_t kSliceShift = 28;  // 2^28 positions = 64 MB of interleaved planes
        const uint64_t positions = genome->dev_words * 32;
        bool scheduled = m >= (1u << 22) && positions > (3ull << kSliceShift);
        if (ctx->dbg.score_slices >= 0) scheduled = ctx->dbg.score_slices == 1;  // tests / experiments
        if (ctx->dbg.score_slice_shift) kSliceShift = std::min<uint32_t>(31, std::max<uint32_t>(8, ctx->dbg.score_slice_shift));
        if (scheduled) {
            uint32_t g_edge[2] = {0, 0};
            VSC_HIP(ctx, hipMemcpyAsync(&g_edge[0], &s.hits[0].guide, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            VSC_HIP(ctx, hipMemcpyAsync(&g_edge[1], &s.hits[m - 1].guide, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const uint32_t n_slices = (uint32_t)((positions + (1ull << kSliceShift) - 1) >> kSliceShift);
            const uint64_t n_pairs = 2ull * (g_edge[1] - g_edge[0] + 1), n_segs = n_pairs * n_slices;
            if (g_edge[1] >= g_edge[0] && n_segs < (1u << 24)) {
                const size_t words = (size_t)(n_pairs * (n_slices + 1) + n_segs + n_segs + 1);
                VSC_HIP(ctx, ctx->score_sched.ensure(words * sizeof(uint64_t)));
                uint64_t *bounds = (uint64_t *)ctx->score_sched.p;
                uint64_t *seg_start = bounds + n_pairs * (n_slices + 1), *seg_prefix = seg_start + n_segs;
                VSC_HIP(ctx, launch_score_schedule(s, g_edge[0], g_edge[1], kSliceShift, n_slices, bounds, seg_start, seg_prefix, ctx->stream));
                s.seg_start = seg_start;
                s.seg_prefix = seg_prefix;
                s.n_segs = (uint32_t)n_segs;
            }
        }
        VSC_HIP(ctx, launch_score_packed(s, dst, ctx->stream));
        VSC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        if (packed_host)
            VSC_HIP(ctx, hipMemcpyAsync(packed_host + done * 16, dst, m * VSC_PACKED_FEATURE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        if (mit_host) VSC_HIP(ctx, hipMemcpyAsync(mit_host + done, s.mit, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        VSC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        total_ms += ms;
    }
    ctx->timing.score_ms = total_ms;
    return VSC_OK;
    });
}

}  // extern "C"

namespace {

// The forest on the device (vsc_rf_predict*, vsc_score_classify_hits): 8-byte integer nodes over the forest's own
// compact column numbering + the sorted distinct activity thresholds.  Kept on the context until a call brings
// another forest (fingerprint of the model's arrays): a streamed search classifies batch after batch.
// Where column `col` (0..441; 442 = the activity rank) sits in the row the kernel extracts tests from
// (rf_row_words in vsc_kernels.hip; the packed layout is feature_row_packed's)
void row_field(uint32_t col, uint8_t *word, uint8_t *shift, uint8_t *width)
{
    auto set = [&](uint32_t w, uint32_t s, uint32_t n) {
        *word = (uint8_t)w;
        *shift = (uint8_t)s;
        *width = (uint8_t)n;
    };
    if (col == VSC_N_FEATURES) return set(19, 0, 8);          // activity rank
    if (col == 0) return set(0, 21, 5);                        // totalMismatches
    if (col <= 21) return set(0, col - 1, 1);                  // mismatchPos1..21
    if (col <= 33) return set(1, col - 22, 1);                 // AtoC..TtoG
    if (col == 34) return set(1, 12, 5);                       // transitions
    if (col == 35) return set(1, 17, 5);                       // transversions
    if (col < 120) return set(2 + ((col - 36) >> 5), (col - 36) & 31u, 1);    // A1..T20, PAMA..PAMT
    if (col < 424) return set(5 + ((col - 120) >> 5), (col - 120) & 31u, 1);  // AA1..TT19
    if (col < 440) return set(16 + (col - 424) / 6, 5 * ((col - 424) % 6), 5);  // AA..TT (dinucleotide counts)
    if (col == 440) return set(0, 26, 5);                      // adjacentMismatches
    return set(1, 22, 4);                                      // seedMismatches
}

int prepare_forest(vsc_ctx *ctx, const vsc_rf_model *model, const char *who)
{
    if (!model || !model->node_status || !model->feature || !model->left || !model->right || !model->split ||
        !model->node_class || model->n_trees == 0 || model->n_nodes == 0)
        return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": null or empty forest").c_str());
    const size_t nn = (size_t)model->n_trees * model->n_nodes;
    if (model->n_nodes > (uint32_t)kRfMaxNodes || (size_t)model->n_nodes * sizeof(uint32_t) > (size_t)kRfTileBytes)
        return fail(ctx, VSC_ERR_RANGE, (std::string(who) + ": a tree has more nodes than the kernel stages at once").c_str());
    if (model->n_trees > 65535u) return fail(ctx, VSC_ERR_RANGE, (std::string(who) + ": more than 65 535 trees").c_str());
    uint64_t h = 0xcbf29ce484222325ull ^ ((uint64_t)model->n_trees << 32 | model->n_nodes);
    h = hash_words(model->node_status, nn, h);
    h = hash_words(model->feature, nn * 2, h);
    h = hash_words(model->left, nn * 2, h);
    h = hash_words(model->right, nn * 2, h);
    h = hash_words(model->split, nn * 8, h);
    h = hash_words(model->node_class, nn, h);
    vsc_ctx::Forest &f = ctx->forest;
    const int want_form = ctx->dbg.rf_form;  // (test hook: -1 = the best form the forest allows)
    h ^= (uint64_t)(uint32_t)(want_form + 2) * 0x9E3779B97F4A7C15ull;
    if (f.nodes.p && f.fingerprint == h && f.n_trees == model->n_trees && f.n_nodes == model->n_nodes) return VSC_OK;
    f.fingerprint = 0;
    // distinct activity thresholds, ascending: `activity <= T_j` <=> `rank(activity) <= j`
    std::vector<double> thr;
    for (size_t i = 0; i < nn; ++i)
        if (model->node_status[i] == 1 && model->feature[i] == VSC_N_FEATURES) thr.push_back(model->split[i]);
    std::sort(thr.begin(), thr.end());
    thr.erase(std::unique(thr.begin(), thr.end()), thr.end());
    if (thr.size() > 255) return fail(ctx, VSC_ERR_RANGE, (std::string(who) + ": the forest splits the on-target activity at more than 255 values").c_str());
    // the distinct tests (column, integer threshold); a split below zero is never met, one at or above 255 always
    struct Split { uint16_t col; uint8_t thr; bool never; };
    auto split_of = [&](size_t i) {
        Split sp{model->feature[i], 0, false};
        const double v = model->split[i];
        if (sp.col == VSC_N_FEATURES) sp.thr = (uint8_t)(std::lower_bound(thr.begin(), thr.end(), v) - thr.begin());
        else if (v < 0) sp.never = true;
        else sp.thr = (uint8_t)std::min(255.0, std::floor(v));
        return sp;
    };
    std::vector<uint32_t> keys;  // col << 8 | thr of every split node that can go either way
    for (size_t i = 0; i < nn; ++i) {
        if (model->node_status[i] != 1) continue;
        const uint16_t ft = model->feature[i];
        // randomForest numbers the daughters of a node behind it: a daughter at or before its parent is a cycle
        const uint32_t own = (uint32_t)(i % model->n_nodes) + 1;  // 1-based index of this node in its tree
        if (ft > VSC_N_FEATURES || model->left[i] <= own || model->right[i] <= own || model->left[i] > model->n_nodes ||
            model->right[i] > model->n_nodes)
            return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": malformed forest (feature or daughter index out of range, or a daughter that does not lie behind its parent)").c_str());
        if (!(model->split[i] == model->split[i])) return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": malformed forest (NaN split)").c_str());
        const Split sp = split_of(i);
        if (!sp.never) keys.push_back((uint32_t)sp.col << 8 | sp.thr);
    }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (keys.empty()) keys.push_back(0);  // (a forest of stumps without a usable split still needs a test table)
    if (keys.size() > (size_t)kRfMaxTests)
        return fail(ctx, VSC_ERR_RANGE, (std::string(who) + ": the forest uses more than 1 024 distinct (predictor, threshold) tests").c_str());
    // tests sorted by the row word they read (the kernel walks the words in a compile-time loop)
    std::vector<RfTest> tests(keys.size());
    for (size_t i = 0; i < keys.size(); ++i) {
        RfTest &ts = tests[i];
        ts.dense_col = (uint16_t)(keys[i] >> 8);
        ts.thr = (uint8_t)keys[i];
        ts.pad = 0;
        row_field(ts.dense_col, &ts.word, &ts.shift, &ts.width);
    }
    std::stable_sort(tests.begin(), tests.end(), [](const RfTest &x, const RfTest &y) { return x.word < y.word; });
    std::vector<uint32_t> test_begin(kRfRowWords + 1, 0);
    for (const RfTest &ts : tests) test_begin[ts.word + 1]++;
    for (int k = 0; k < kRfRowWords; ++k) test_begin[k + 1] += test_begin[k];
    auto test_index = [&](uint16_t col, uint8_t t) {
        for (size_t i = 0; i < tests.size(); ++i)
            if (tests[i].dense_col == col && tests[i].thr == t) return (uint32_t)i;
        return 0u;
    };
    std::vector<uint32_t> index_of(keys.size());  // by position in `keys`
    for (size_t i = 0; i < keys.size(); ++i) index_of[i] = test_index((uint16_t)(keys[i] >> 8), (uint8_t)keys[i]);
    // nodes: test | left << 10 | right << 20 (0-based) | terminal << 30 | votes class "1" << 31; or the compact form
    // (trees of <= 512 nodes): test | (nodes to skip | vote << 10) << 10 (right) / << 21 (left)
    const bool compact = model->n_nodes <= 512 && want_form != 0;
    std::vector<uint32_t> nodes(nn);
    std::vector<uint8_t> depth(model->n_trees, 0);
    std::vector<uint16_t> level(model->n_nodes);
    for (uint32_t tr = 0; tr < model->n_trees; ++tr) {
        std::fill(level.begin(), level.end(), 0);
        for (uint32_t k = 0; k < model->n_nodes; ++k) {
            const size_t i = (size_t)tr * model->n_nodes + k;
            if (model->node_status[i] != 1) {  // terminal (or an unused slot behind the tree's last node: never reached)
                nodes[i] = k << 10 | k << 20 | 1u << 30 | (model->node_class[i] == 2 ? 1u << 31 : 0u);
                continue;
            }
            const Split sp = split_of(i);
            uint32_t l = model->left[i] - 1u, r = model->right[i] - 1u, test = 0;
            if (sp.never) l = r;  // x <= (negative) never holds: both ways lead right
            else test = index_of[std::lower_bound(keys.begin(), keys.end(), (uint32_t)sp.col << 8 | sp.thr) - keys.begin()];
            nodes[i] = test | l << 10 | r << 20;
            if (compact) {
                // (x <= thr takes the LEFT daughter: filed in the upper field, which the kernel selects with 10 + 11 * bit)
                // a daughter field = how many nodes further on the walk continues (10 bits) | vote (bit 10): a split
                // daughter lies d - k nodes behind its parent; a terminal one sends the walk to the NEXT tree's root,
                // n_nodes - k nodes on (trees lie back to back at a stride of n_nodes), and brings its vote along
                auto daughter = [&](uint32_t d) {
                    const size_t j = (size_t)tr * model->n_nodes + d;
                    return model->node_status[j] == 1 ? d - k : ((model->n_nodes - k) | (model->node_class[j] == 2 ? 1u << 10 : 0u));
                };
                nodes[i] = test | daughter(r) << 10 | daughter(l) << 21;
            }
            level[l] = level[r] = (uint16_t)(level[k] + 1);  // (daughters lie behind their parent: level[k] is final here)
            depth[tr] = (uint8_t)std::min<uint32_t>(255, std::max<uint32_t>(depth[tr], level[k] + 1u));
        }
        if (compact && model->node_status[(size_t)tr * model->n_nodes] != 1) {
            // a tree that is one terminal node: a root whose daughters both are "terminal, the root's vote"
            const uint32_t leaf = model->n_nodes | (model->node_class[(size_t)tr * model->n_nodes] == 2 ? 1u << 10 : 0u);
            nodes[(size_t)tr * model->n_nodes] = leaf << 10 | leaf << 21;
        }
    }
    // PAIR form (vsc_internal.h): a node of 8 bytes holds a split node AND its two daughters - two levels per LDS read.
    // Pair nodes are rooted at the split nodes on even levels; a tree's pair nodes lie in breadth-first order (every exit
    // leads forward), the trees back to back at a stride of `pair_stride` nodes.  For forests of at most 232 tests and 127
    // pair nodes per tree (rfClassifier: 217 and 70).
    std::vector<uint64_t> pairs;
    uint32_t pair_stride = 0;
    if (compact && want_form != 1 && tests.size() <= (size_t)kRfPairMaxTests) {
        auto split = [&](uint32_t tr, uint32_t k) { return model->node_status[(size_t)tr * model->n_nodes + k] == 1; };
        std::vector<std::vector<uint32_t>> roots(model->n_trees);  // per tree: the split nodes that root a pair node, in order
        std::vector<uint32_t> index_in(model->n_nodes);
        // (first pass: the roots of every tree; second pass below fills the words once the stride is known)
        auto daughters = [&](uint32_t tr, uint32_t k, uint32_t &l, uint32_t &r, uint32_t &test) {
            const size_t i = (size_t)tr * model->n_nodes + k;
            const Split sp = split_of(i);
            l = model->left[i] - 1u, r = model->right[i] - 1u, test = 0;
            if (sp.never) l = r;
            else test = index_of[std::lower_bound(keys.begin(), keys.end(), (uint32_t)sp.col << 8 | sp.thr) - keys.begin()];
        };
        for (uint32_t tr = 0; tr < model->n_trees; ++tr) {
            auto &q = roots[tr];
            q.push_back(0);  // (a tree that is one terminal node: a pair node whose four exits carry the root's vote)
            for (size_t at = 0; at < q.size() && split(tr, q[at]); ++at) {
                uint32_t l, r, test;
                daughters(tr, q[at], l, r, test);
                for (uint32_t c : {l, r}) {
                    if (!split(tr, c)) continue;
                    uint32_t cl, cr, ct;
                    daughters(tr, c, cl, cr, ct);
                    for (uint32_t g : {cl, cr})
                        if (split(tr, g) && std::find(q.begin(), q.end(), g) == q.end()) q.push_back(g);
                }
            }
            pair_stride = std::max<uint32_t>(pair_stride, (uint32_t)q.size());
        }
        if (pair_stride <= 127) {
            pairs.assign((size_t)model->n_trees * pair_stride, 0);
            for (uint32_t tr = 0; tr < model->n_trees; ++tr) {
                const auto &q = roots[tr];
                for (uint32_t j = 0; j < q.size(); ++j) index_in[q[j]] = j;
                auto vote_of = [&](uint32_t k) { return model->node_class[(size_t)tr * model->n_nodes + k] == 2 ? 1u : 0u; };
                for (uint32_t j = 0; j < q.size(); ++j) {
                    // an exit (one byte): vote | pair nodes to skip << 1 - to the pair node rooted at a split granddaughter, or -
                    // terminal - to the next tree's root with the vote
                    auto exit_to = [&](uint32_t g) { return split(tr, g) ? (index_in[g] - j) << 1 : ((pair_stride - j) << 1 | vote_of(g)); };
                    uint32_t ex[4], t_root = 0, t_left = 0, t_right = 0;  // ex[2 * (root bit) + (daughter bit)]; bit set = x <= thr = LEFT
                    if (!split(tr, q[j])) {
                        ex[0] = ex[1] = ex[2] = ex[3] = exit_to(q[j]);
                    } else {
                        uint32_t l, r;
                        daughters(tr, q[j], l, r, t_root);
                        auto half = [&](uint32_t c, uint32_t &t, uint32_t &on_left, uint32_t &on_right) {
                            if (!split(tr, c)) { t = 0; on_left = on_right = exit_to(c); return; }
                            uint32_t cl, cr;
                            daughters(tr, c, cl, cr, t);
                            on_left = exit_to(cl), on_right = exit_to(cr);
                        };
                        half(l, t_left, ex[3], ex[2]);
                        half(r, t_right, ex[1], ex[0]);
                    }
                    // a test in the upper word: its bit in the row's test word at bits s .. s + 4, the word's number at s + 10 ..
                    // s + 12 (s = 0 root, 5 right daughter, 18 left daughter: rf_predict_kernel masks the word number in place)
                    // (test i: word i / 29, bit 3 + i % 29; the root's field holds the bit's position, a daughter's the position - 3)
                    const uint32_t per = 32u - (uint32_t)kRfPairFirstBit;
                    auto field = [&](uint32_t test, uint32_t s, bool root) {
                        return (uint64_t)((test % per + (root ? (uint32_t)kRfPairFirstBit : 0u)) | (test / per) << 10) << s;
                    };
                    uint64_t w = 0;
                    for (int e = 0; e < 4; ++e) w |= (uint64_t)ex[e] << (8 * e);
                    w |= (field(t_root, 0, true) | field(t_right, 5, false) | field(t_left, 18, false)) << 32;
                    pairs[(size_t)tr * pair_stride + j] = w;
                }
            }
        }
    }
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    auto pad = [](size_t n) { return (n + 255) / 256 * 256; };
    const size_t node_bytes = pairs.empty() ? nn * sizeof(uint32_t) : pairs.size() * sizeof(uint64_t);
    const size_t nodes_b = pad(node_bytes), depth_b = pad(depth.size()), tests_b = pad(tests.size() * sizeof(RfTest));
    VSC_HIP(ctx, f.nodes.ensure(nodes_b + depth_b + tests_b + pad(test_begin.size() * sizeof(uint32_t))));
    char *base = (char *)f.nodes.p;
    VSC_HIP(ctx, hipMemcpyAsync(base, pairs.empty() ? (const void *)nodes.data() : (const void *)pairs.data(), node_bytes, hipMemcpyHostToDevice, ctx->stream));
    VSC_HIP(ctx, hipMemcpyAsync(base + nodes_b, depth.data(), depth.size(), hipMemcpyHostToDevice, ctx->stream));
    VSC_HIP(ctx, hipMemcpyAsync(base + nodes_b + depth_b, tests.data(), tests.size() * sizeof(RfTest), hipMemcpyHostToDevice, ctx->stream));
    VSC_HIP(ctx, hipMemcpyAsync(base + nodes_b + depth_b + tests_b, test_begin.data(), test_begin.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the host vectors go out of scope)
    f.depth_at = nodes_b;
    f.tests_at = nodes_b + depth_b;
    f.begin_at = nodes_b + depth_b + tests_b;
    f.n_tests = (uint32_t)tests.size();
    f.form = !pairs.empty() ? 2u : compact ? 1u : 0u;
    f.node_stride = !pairs.empty() ? pair_stride : model->n_nodes;
    f.n_trees = model->n_trees;
    f.n_nodes = model->n_nodes;
    f.thresholds = thr;
    f.fingerprint = h;
    return VSC_OK;
}

void fill_forest(RfArgs &a, const vsc_ctx *ctx)
{
    const vsc_ctx::Forest &f = ctx->forest;
    const char *base = (const char *)f.nodes.p;
    a.nodes = (const uint32_t *)base;
    a.depth = (const uint8_t *)(base + f.depth_at);
    a.tests = (const RfTest *)(base + f.tests_at);
    a.test_begin = (const uint32_t *)(base + f.begin_at);
    a.n_tests = f.n_tests;
    a.compact = f.form;
    a.n_trees = f.n_trees;
    a.n_nodes = f.node_stride;
}

uint8_t activity_rank(const std::vector<double> &thr, double activity)
{
    return (uint8_t)(std::lower_bound(thr.begin(), thr.end(), activity) - thr.begin());  // thresholds strictly below
}

// vsc_rf_predict / vsc_rf_predict_packed: the rows from the host (dense) or from host / device memory (packed)
int rf_predict(vsc_ctx *ctx, const vsc_rf_model *model, const uint8_t *dense, const void *packed, int packed_on_device,
               const double *activity, uint64_t n, double *prob, uint8_t *cls, uint8_t *tie, const char *who)
{
    ctx->err.clear();
    if (n && ((!dense && !packed) || !activity)) return fail(ctx, VSC_ERR_INVALID, (std::string(who) + ": null or empty argument").c_str());
    const int frc = prepare_forest(ctx, model, who);
    if (frc != VSC_OK) return frc;
    if (n == 0) return VSC_OK;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint8_t> ranks(n);
    for (uint64_t i = 0; i < n; ++i) ranks[i] = activity_rank(ctx->forest.thresholds, activity[i]);
    VSC_HIP(ctx, ctx->score_mit.ensure(n));
    VSC_HIP(ctx, ctx->score_flags.ensure(n * sizeof(uint32_t)));
    VSC_HIP(ctx, hipMemcpyAsync(ctx->score_mit.p, ranks.data(), n, hipMemcpyHostToDevice, ctx->stream));
    RfArgs a{};
    fill_forest(a, ctx);
    a.act_rank = (const uint8_t *)ctx->score_mit.p;
    a.n = n;
    a.votes = (uint32_t *)ctx->score_flags.p;
    if (dense) {
        VSC_HIP(ctx, ctx->score_feat.ensure(n * VSC_N_FEATURES));
        VSC_HIP(ctx, hipMemcpyAsync(ctx->score_feat.p, dense, n * VSC_N_FEATURES, hipMemcpyHostToDevice, ctx->stream));
        a.dense = (const uint8_t *)ctx->score_feat.p;
    } else if (packed_on_device) {
        a.packed = (const uint4 *)packed;
    } else {
        VSC_HIP(ctx, ctx->score_feat.ensure(n * VSC_PACKED_FEATURE_BYTES));
        VSC_HIP(ctx, hipMemcpyAsync(ctx->score_feat.p, packed, n * VSC_PACKED_FEATURE_BYTES, hipMemcpyHostToDevice, ctx->stream));
        a.packed = (const uint4 *)ctx->score_feat.p;
    }
    // few rows: split the trees over several workgroups per row tile so that the device is filled
    const uint64_t tiles = (n + kRfRows - 1) / kRfRows;
    a.tree_splits = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)2 * ctx->n_cus / tiles, 32, model->n_trees}));
    if (a.tree_splits > 1) VSC_HIP(ctx, hipMemsetAsync(a.votes, 0, n * sizeof(uint32_t), ctx->stream));
    VSC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    VSC_HIP(ctx, launch_rf_predict(a, ctx->stream));
    VSC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    std::vector<uint32_t> votes(n);
    VSC_HIP(ctx, hipMemcpyAsync(votes.data(), a.votes, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t i = 0; i < n; ++i) {
        if (prob) prob[i] = (double)votes[i] / (double)model->n_trees;  // type = "prob": votes / ntree
        if (cls) cls[i] = 2 * votes[i] > model->n_trees;
        if (tie) tie[i] = 2 * votes[i] == model->n_trees;
    }
    float ms = 0;
    VSC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timing.score_ms = ms;
    return VSC_OK;
}

}  // namespace

extern "C" {

int vsc_rf_predict(vsc_ctx *ctx, const vsc_rf_model *model, const uint8_t *features, const double *activity, uint64_t n,
                   double *prob, uint8_t *cls, uint8_t *tie)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    return rf_predict(ctx, model, features, nullptr, 0, activity, n, prob, cls, tie, "vsc_rf_predict");
    });
}

int vsc_rf_predict_packed(vsc_ctx *ctx, const vsc_rf_model *model, const void *packed_rows, int rows_on_device,
                          const double *activity, uint64_t n, double *prob, uint8_t *cls, uint8_t *tie)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    return rf_predict(ctx, model, nullptr, packed_rows, rows_on_device, activity, n, prob, cls, tie, "vsc_rf_predict_packed");
    });
}

int vsc_score_classify_hits(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, const uint64_t *guides, uint32_t n_guides,
                            const double *guide_activity, const vsc_rf_model *model, uint64_t first, uint64_t count, void *votes_dev,
                            uint16_t *votes_host, double *mit_host)
{
    return guarded(ctx, [&]() -> int {
    if (!ctx) return VSC_ERR_INVALID;
    ctx->err.clear();
    if (!genome || !hits || (n_guides && (!guides || !guide_activity)))
        return fail(ctx, VSC_ERR_INVALID, "vsc_score_classify_hits: null argument");
    if (first > hits->n || count > hits->n - first) return fail(ctx, VSC_ERR_INVALID, "vsc_score_classify_hits: row range outside the result");
    const int frc = prepare_forest(ctx, model, "vsc_score_classify_hits");
    if (frc != VSC_OK) return frc;
    ctx->timing.score_ms = 0;
    if (count == 0) return VSC_OK;
    VSC_HIP(ctx, hipSetDevice(ctx->device));
    VSC_HIP(ctx, upload_read_planes(ctx, guides, n_guides));
    VSC_HIP(ctx, ensure_hl(ctx, genome));
    // the reads' activity ranks: uploaded when the activities or the forest differ from the last call's (a streamed search
    // classifies batch after batch with the same ones)
    const uint64_t rank_key = hash_words(guide_activity, (size_t)n_guides * sizeof(double), ctx->forest.fingerprint ^ n_guides);
    if (!ctx->forest.ranks.p || ctx->forest.ranks_key != rank_key || ctx->forest.ranks_n != n_guides) {
        std::vector<uint8_t> ranks(std::max<uint32_t>(n_guides, 1));
        for (uint32_t g = 0; g < n_guides; ++g) ranks[g] = activity_rank(ctx->forest.thresholds, guide_activity[g]);
        ctx->forest.ranks_n = ~0u;
        VSC_HIP(ctx, ctx->forest.ranks.ensure(ranks.size()));
        VSC_HIP(ctx, hipMemcpyAsync(ctx->forest.ranks.p, ranks.data(), ranks.size(), hipMemcpyHostToDevice, ctx->stream));
        VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `ranks` goes out of scope
        ctx->forest.ranks_key = rank_key;
        ctx->forest.ranks_n = n_guides;
    }
    // 2 bytes (+ 8 with the MIT score) per hit of scratch: one pass for any result that fits the device at all
    uint64_t rows = 0;
    VSC_HIP(ctx, score_scratch(ctx, count, mit_host ? sizeof(double) : 0, votes_dev ? 0 : sizeof(uint16_t), 0, &rows));
    double total_ms = 0;
    for (uint64_t done = 0; done < count; done += rows) {
        const uint64_t m = std::min(rows, count - done);
        RfArgs a{};
        fill_forest(a, ctx);
        fill_score_args(a.score, ctx, genome);
        a.score.hits = hits->d_records + first + done;
        a.score.n = m;
        if (mit_host) a.score.mit = (double *)ctx->score_mit.p;
        a.act_rank = (const uint8_t *)ctx->forest.ranks.p;
        a.n = m;
        a.votes16 = votes_dev ? (uint16_t *)votes_dev + done : (uint16_t *)ctx->score_flags.p;
        a.tree_splits = 1;
        VSC_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        VSC_HIP(ctx, launch_rf_predict(a, ctx->stream));
        VSC_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        if (votes_host) VSC_HIP(ctx, hipMemcpyAsync(votes_host + done, a.votes16, m * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
        if (mit_host) VSC_HIP(ctx, hipMemcpyAsync(mit_host + done, a.score.mit, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        VSC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        VSC_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        total_ms += ms;
    }
    ctx->timing.score_ms = total_ms;
    return VSC_OK;
    });
}

}  // extern "C"
