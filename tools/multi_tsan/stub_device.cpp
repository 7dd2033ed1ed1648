// stub_device.cpp - a HOST stand-in for everything below csrc/vsc_multi.cpp, so that the multi-device engine's threads (a
// host thread per shard, two exchange slots, the calling thread sending / merging / calling back) can run under
// ThreadSanitizer on a machine without a GPU (GPU sanitizers are not available on this pool).
// TEST INFRASTRUCTURE ONLY: nothing here is built into libvarscot_hip.so or the tools (tools/multi_tsan/run.sh links it with
// vsc_multi.cpp into a test program of its own).
//   * "device memory" is host memory; hipMemcpyAsync / hipMemcpyPeerAsync only QUEUE their copy on the stream, and
//     hipStreamSynchronize performs the queue - a copy happens as late as the real one may, so an exchange slot handed back
//     to its shard before the synchronisation shows up as a race (or as wrong records in the merge below);
//   * a "shard search" invents a deterministic result from (read code, strand, shard): fake_count() hits with ascending
//     positions inside the shard's range, after a short pseudo-random sleep that shuffles the threads' interleavings;
//   * the exchange record is key << 40 | position, so that the fake merge can CHECK every record it places (right key, right
//     shard order, ascending positions) - a slot overwritten too early or a count from another batch fails there.
#include <chrono>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "stub_device.h"
#include "vsc_objects.h"

namespace stub {
std::atomic<int> fail_shard{-1};            // the shard (context serial within its vsc_multi) whose search fails ...
std::atomic<uint64_t> fail_code{~0ull};     // ... on the batch whose first read has this code
std::atomic<uint64_t> merges{0}, copies{0};
std::atomic<int> ctx_serial{0};             // (the driver resets it before vsc_multi_create: context i of a set is shard i)

uint64_t mix(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    return x ^ (x >> 33);
}
uint32_t fake_count(uint64_t code, uint32_t strand, uint64_t shard_first_word) { return (uint32_t)(mix(code * 2 + strand + shard_first_word * 0x9E3779B97F4A7C15ull) % 6); }
uint16_t fake_vote(uint64_t code, uint32_t strand, uint32_t pos) { return (uint16_t)(mix(code ^ ((uint64_t)pos << 20) ^ strand) & 0x3FF); }
}  // namespace stub

// ---- the HIP runtime entry points vsc_multi.cpp and DeviceBuf use -------------------------------------------------------------
namespace {
struct Copy {
    void *dst;
    const void *src;
    size_t n;
};
struct Stream {
    std::mutex mu;
    std::deque<Copy> queue;
};
thread_local int tl_device = 0;
}  // namespace

extern "C" {
hipError_t hipSetDevice(int d)
{
    tl_device = d;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub error"; }
hipError_t hipMalloc(void **p, size_t n)
{
    *p = ::operator new(n ? n : 1, std::nothrow);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p)
{
    ::operator delete(p);
    return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s)
{
    *s = (hipStream_t) new Stream();
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    delete (Stream *)s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    Stream *st = (Stream *)s;
    std::deque<Copy> q;
    {
        std::lock_guard<std::mutex> lk(st->mu);
        q.swap(st->queue);
    }
    for (const Copy &c : q) std::memcpy(c.dst, c.src, c.n);
    stub::copies += q.size();
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s)
{
    Stream *st = (Stream *)s;
    std::lock_guard<std::mutex> lk(st->mu);
    st->queue.push_back(Copy{dst, src, n});
    return hipSuccess;
}
hipError_t hipMemcpyPeerAsync(void *dst, int, const void *src, int, size_t n, hipStream_t s) { return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s); }

// ---- the single-device entry points of the library that vsc_multi.cpp calls ---------------------------------------------------
int vsc_ctx_create(int device, vsc_ctx **out)
{
    *out = new vsc_ctx();
    (*out)->device = device;
    (*out)->n_cus = stub::ctx_serial++;  // (this field is free here: the stub's serial number of the context)
    return VSC_OK;
}
int vsc_ctx_destroy(vsc_ctx *c)
{
    delete c;
    return VSC_OK;
}
int vsc_ctx_release_scratch(vsc_ctx *) { return VSC_OK; }
int vsc_ctx_timing(const vsc_ctx *c, vsc_timing *out)
{
    *out = c->timing;
    return VSC_OK;
}
const char *vsc_last_error(const vsc_ctx *c) { return c ? c->err.c_str() : "null context"; }

int vsc_genome_load(vsc_ctx *ctx, const uint32_t *, const uint32_t *, const uint32_t *, uint64_t first_word, uint64_t n_words, uint64_t own_words,
                    const vsc_contig *, uint32_t n_contigs, vsc_genome **out)
{
    vsc_genome *g = new vsc_genome();
    g->ctx = ctx;
    g->first_word = first_word;
    g->own_words = own_words;
    g->dev_words = n_words;
    g->n_contigs = n_contigs;
    *out = g;
    return VSC_OK;
}
int vsc_genome_free(vsc_genome *g)
{
    delete g;
    return VSC_OK;
}
int vsc_genome_build_index(vsc_ctx *, vsc_genome *g, const vsc_search_params *)
{
    g->has_index = true;
    return VSC_OK;
}

uint64_t vsc_hits_count(const vsc_hits *h) { return h ? h->n : 0; }
int vsc_hits_free(vsc_hits *h)
{
    delete h;
    return VSC_OK;
}

int vsc_search(vsc_ctx *ctx, const vsc_genome *g, const uint64_t *guides, uint32_t n_guides, const vsc_search_params *, vsc_hits **out)
{
    *out = nullptr;
    const uint64_t seed = stub::mix(g->first_word ^ (n_guides ? guides[0] : 0));
    std::this_thread::sleep_for(std::chrono::microseconds(seed % 700));
    if (ctx->n_cus == stub::fail_shard.load() && n_guides && guides[0] == stub::fail_code.load()) {
        ctx->err = "stub: this shard was told to fail";
        return VSC_ERR_DEVICE;
    }
    vsc_hits *h = new vsc_hits();
    h->ctx = ctx;
    for (uint32_t i = 0; i < n_guides; ++i)
        for (uint32_t s = 0; s < 2; ++s) {
            const uint32_t c = stub::fake_count(guides[i], s, g->first_word);
            for (uint32_t j = 0; j < c; ++j) {
                vsc_hit r{};
                r.guide = i;
                r.contig = 0;
                r.pos = (uint32_t)(g->first_word * 32 + 7 * j + (uint32_t)(stub::mix(guides[i] + s) % 5));
                r.info = s << 31;
                h->host.push_back(r);
            }
        }
    h->n = h->host.size();
    h->host_valid = true;
    ctx->timing = vsc_timing{};
    ctx->timing.total_ms = 1.0;
    ctx->timing.hits = h->n;
    *out = h;
    return VSC_OK;
}

int vsc_search_stream_rows(vsc_ctx *ctx, const vsc_genome *g, const uint64_t *guides, uint32_t n_guides, const vsc_search_params *p, uint32_t,
                           vsc_rows_batch_fn on_batch, void *user)
{
    if (n_guides == 0) return VSC_OK;
    vsc_hits *h = nullptr;
    int rc = vsc_search(ctx, g, guides, n_guides, p, &h);
    if (rc != VSC_OK) return rc;
    rc = on_batch(user, h, 0, n_guides, nullptr);
    vsc_hits_free(h);
    return rc;
}

int vsc_score_hits_packed(vsc_ctx *ctx, const vsc_genome *, const vsc_hits *, const uint64_t *, uint32_t, uint64_t, uint64_t, void *, uint32_t *, double *)
{
    ctx->timing.score_ms = 0.5;
    return VSC_OK;
}

int vsc_score_classify_hits(vsc_ctx *ctx, const vsc_genome *, const vsc_hits *h, const uint64_t *guides, uint32_t, const double *, const vsc_rf_model *,
                            uint64_t first, uint64_t count, void *votes_dev, uint16_t *, double *)
{
    uint16_t *v = (uint16_t *)votes_dev;
    for (uint64_t i = 0; i < count; ++i) {
        const vsc_hit &r = h->host[first + i];
        v[i] = stub::fake_vote(guides[r.guide], VSC_HIT_STRAND(r.info), r.pos);
    }
    ctx->timing.score_ms = 0.25;
    return VSC_OK;
}

int vsc_hits_pack_exchange(vsc_ctx *, const vsc_genome *, const vsc_hits *h, uint32_t n_guides, void *records, int, uint32_t *key_counts)
{
    uint64_t *out = (uint64_t *)records;
    std::memset(key_counts, 0, 2 * (size_t)n_guides * sizeof(uint32_t));
    for (uint64_t i = 0; i < h->n; ++i) {
        const vsc_hit &r = h->host[i];
        const uint32_t key = r.guide << 1 | VSC_HIT_STRAND(r.info);
        out[i] = (uint64_t)key << 40 | r.pos;
        key_counts[key]++;
    }
    return VSC_OK;
}
}  // extern "C"

namespace vsc {
bool host_timing_on() { return false; }

int genome_table_only(vsc_ctx *ctx, const vsc_contig *, uint32_t n_contigs, vsc_genome **out)
{
    vsc_genome *g = new vsc_genome();
    g->ctx = ctx;
    g->n_contigs = n_contigs;
    *out = g;
    return VSC_OK;
}

int merge_packed_shards(vsc_ctx *ctx, const vsc_genome *, const void *const *shard_records, const void *const *shard_side, const uint32_t *key_counts,
                        uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out, DeviceBuf *side_out)
{
    *out = nullptr;
    vsc_hits *h = new vsc_hits();
    h->ctx = ctx;
    std::vector<uint64_t> at(n_shards, 0);
    std::vector<uint16_t> votes;
    for (uint32_t k = 0; k < n_keys; ++k) {
        uint32_t last_pos = 0;
        bool any = false;
        for (uint32_t s = 0; s < n_shards; ++s)
            for (uint32_t j = 0; j < key_counts[(size_t)s * n_keys + k]; ++j) {
                const uint64_t rec = ((const uint64_t *)shard_records[s])[at[s]];
                if ((uint32_t)(rec >> 40) != k || (any && (uint32_t)rec <= last_pos)) {
                    ctx->err = "stub merge: a record that does not belong here (key " + std::to_string(rec >> 40) + " at key " + std::to_string(k) +
                               ", shard " + std::to_string(s) + ")";
                    delete h;
                    return VSC_ERR_INVALID;
                }
                any = true;
                last_pos = (uint32_t)rec;
                vsc_hit r{};
                r.guide = (first_key + k) >> 1;
                r.pos = (uint32_t)rec;
                r.info = ((first_key + k) & 1u) << 31;
                h->host.push_back(r);
                if (shard_side) votes.push_back(((const uint16_t *)shard_side[s])[at[s]]);
                ++at[s];
            }
    }
    h->n = h->host.size();
    h->host_valid = true;
    if (side_out && !votes.empty()) {
        if (side_out->ensure(votes.size() * sizeof(uint16_t)) != hipSuccess) {
            delete h;
            return VSC_ERR_NOMEM;
        }
        std::memcpy(side_out->p, votes.data(), votes.size() * sizeof(uint16_t));
    }
    stub::merges++;
    *out = h;
    return VSC_OK;
}
}  // namespace vsc
