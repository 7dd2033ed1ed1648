// vsc_multi.cpp - the genome-sharded multi-GPU search behind the C ABI: one host process, one vsc_ctx per
// device, the packed planes cut into tile-aligned position ranges (+ one halo word), every device searching ALL
// reads on its shard from its own host thread, then exactly one exchange of hit records to the first device
// over RCCL (xGMI) and the segment merge there.  This is where VARSCOT_pipeline/read_mapping/bidir_mapping.cpp
// has its OpenMP loop over reads (:285-295) and the concatenation of the per-thread buffers (:307-308): the
// parallel axis is the genome instead of the reads, the concatenation becomes gather + merge.
//
// RCCL is bound at run time (dlopen): the library loads and works on one device without it, and in a Python
// process it shares the copy PyTorch has already loaded.  Devices may repeat in the list (several contexts on
// one GPU - how the tests run N shards on a one-GPU box); the exchange then is plain device copies.
//
// One engine serves vsc_multi_search (one batch) and vsc_multi_search_stream: a host thread per shard searches (and scores)
// batch after batch into one of two exchange slots; the calling thread sends every shard's records to the first device the
// moment THAT shard is ready, merges the batch on a context of its own there and hands it to the caller - while the shards
// are already searching the next batch.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "vsc_objects.h"

using namespace vsc;

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;

    // `only` (tests): the one library name to try instead of the usual two
    bool load(std::string *why, const char *only = nullptr)
    {
        if (lib) return true;
        std::string last;
        std::vector<const char *> names;
        if (only) names = {only};
        else names = {"librccl.so.1", "librccl.so"};
        for (const char *name : names) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
            const char *e = dlerror();  // (one call: dlerror() clears the state it reports)
            last = e ? e : "dlopen failed";
        }
        if (!lib) {
            *why = "RCCL not found: " + last;
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv || !AllGather || !GetErrorString) {
            *why = "RCCL library lacks a required entry point";
            return false;
        }
        return true;
    }
};

}  // namespace

struct vsc_multi {
    std::vector<int> device;
    std::vector<vsc_ctx *> ctx;
    vsc_ctx *merge_ctx = nullptr;       // on device[0]: owns the merged results; its stream is not a shard's
    std::vector<hipStream_t> xstream;   // exchange streams, one per context (RCCL's streams)
    bool use_rccl = false;
    Rccl rccl;
    std::vector<ncclComm_t> comm;
    // exchange buffers: two slots per shard on the shard's device (records, votes) - a shard fills one while the other is
    // being sent - and one landing buffer per shard on device[0]
    std::vector<DeviceBuf> xbuf[2], vbuf[2];
    std::vector<DeviceBuf> gbuf, gvotes;
    DeviceBuf votes_out;                // on device[0]: the merged batch's votes
    std::string err;
    vsc_multi_timing timing{};
};

struct vsc_multi_genome {
    vsc_multi *multi = nullptr;
    std::vector<vsc_genome *> shard;    // null: the shard owns no words of this (small) genome
    vsc_genome *table = nullptr;        // the contig table on the first device, for the merge context
};

namespace {

int mfail(vsc_multi *m, int code, const std::string &what)
{
    if (m) m->err = what;
    return code;
}

// tile-aligned word range of shard r of n (the same cut as varscot_amd.api.PackedGenome.shard_words)
void shard_range(uint64_t n_words, unsigned r, unsigned n, uint64_t *b, uint64_t *e)
{
    const uint64_t tiles = (n_words + kTileWords - 1) / kTileWords;
    *b = std::min<uint64_t>(tiles * r / n * kTileWords, n_words);
    *e = std::min<uint64_t>(tiles * (r + 1) / n * kTileWords, n_words);
}

// joins what was started, whatever ends the scope (a thread that cannot be started, an exception in body(0))
struct JoinAll {
    std::vector<std::thread> &pool;
    ~JoinAll()
    {
        for (auto &t : pool)
            if (t.joinable()) t.join();
    }
};

// body(i) for every i on a thread of its own (i = 0 on this one).  The bodies call the C ABI, which throws nothing.
template <class F> void on_all(size_t n, F &&body)
{
    std::vector<std::thread> pool;
    pool.reserve(n);
    JoinAll join{pool};
    for (size_t i = 1; i < n; ++i) pool.emplace_back([&, i] { body(i); });
    if (n) body(0);
}

}  // namespace

// no C++ exception crosses the C boundary
template <class F> int mguarded(vsc_multi *m, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try {
            if (m) m->err = "out of host memory";
        } catch (...) {
        }
        return VSC_ERR_NOMEM;
    } catch (...) {
        try {
            if (m) m->err = "unexpected C++ exception";
        } catch (...) {
        }
        return VSC_ERR_DEVICE;
    }
}

extern "C" {

int vsc_multi_create(const int *device_ids, int n, vsc_multi **out) { return vsc_multi_create_debug(device_ids, n, nullptr, out); }

int vsc_multi_create_debug(const int *device_ids, int n, const vsc_multi_debug_params *params, vsc_multi **out)
{
    return mguarded(nullptr, [&]() -> int {
    if (!out) return VSC_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n <= 0 || n > 64) return VSC_ERR_INVALID;
    // (held by a smart pointer until it is handed over: an exception half-way releases the contexts and streams made so far)
    std::unique_ptr<vsc_multi, int (*)(vsc_multi *)> holder(new (std::nothrow) vsc_multi(), vsc_multi_destroy);
    vsc_multi *m = holder.get();
    if (!m) return VSC_ERR_NOMEM;
    m->device.assign(device_ids, device_ids + n);
    m->ctx.assign(n, nullptr);
    m->xstream.assign(n, nullptr);
    for (int k = 0; k < 2; ++k) {
        m->xbuf[k].assign(n, DeviceBuf{});
        m->vbuf[k].assign(n, DeviceBuf{});
    }
    m->gbuf.assign(n, DeviceBuf{});
    m->gvotes.assign(n, DeviceBuf{});
    for (int i = 0; i < n; ++i) {
        const int rc = vsc_ctx_create(device_ids[i], &m->ctx[i]);
        if (rc != VSC_OK) return rc;
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreate(&m->xstream[i]) != hipSuccess) return VSC_ERR_DEVICE;
    }
    const int mrc = vsc_ctx_create(device_ids[0], &m->merge_ctx);
    if (mrc != VSC_OK) return mrc;
    // RCCL needs one communicator rank per DISTINCT device; a list with repeats (tests, rehearsals) exchanges by
    // device copies.  Hooks (varscot_hip_debug.h): rccl = 0 forces copies, 1 insists on RCCL (an error if it
    // cannot be set up).
    std::vector<int> sorted(m->device);
    std::sort(sorted.begin(), sorted.end());
    const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    const bool forced = params && params->rccl == 1, off = params && params->rccl == 0, attempt = params && params->rccl == 2;
    if (distinct && !off && (n > 1 || forced || attempt)) {
        std::string why;
        bool ok = m->rccl.load(&why, params ? params->rccl_library : nullptr);
        if (ok) {
            m->comm.assign(n, nullptr);
            const ncclResult_t r = m->rccl.CommInitAll(m->comm.data(), n, m->device.data());
            if (r != ncclSuccess) {
                ok = false;
                why = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r);
                m->comm.clear();
            }
        }
        if (!ok && forced) {
            std::fprintf(stderr, "vsc_multi_create: %s\n", why.c_str());
            return VSC_ERR_DEVICE;
        }
        m->use_rccl = ok;  // not forced and unavailable: peer copies carry the records instead
        if (!ok) m->err = why;  // (vsc_multi_last_error says why the copies are in use)
    } else if (forced) {
        std::fprintf(stderr, "vsc_multi_create: RCCL was asked for but needs distinct devices\n");
        return VSC_ERR_INVALID;
    }
    *out = holder.release();
    return VSC_OK;
    });
}

int vsc_multi_destroy(vsc_multi *m)
{
    if (!m) return VSC_OK;
    if (!m->comm.empty())
        for (ncclComm_t c : m->comm)
            if (c) (void)m->rccl.CommDestroy(c);
    if (!m->device.empty()) {
        (void)hipSetDevice(m->device[0]);
        for (auto &b : m->gbuf) b.release();
        for (auto &b : m->gvotes) b.release();
        m->votes_out.release();
        if (m->merge_ctx) vsc_ctx_destroy(m->merge_ctx);
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void)hipSetDevice(m->device[i]);
        for (int k = 0; k < 2; ++k) {
            if (i < m->xbuf[k].size()) m->xbuf[k][i].release();
            if (i < m->vbuf[k].size()) m->vbuf[k][i].release();
        }
        if (i < m->xstream.size() && m->xstream[i]) (void)hipStreamDestroy(m->xstream[i]);
        if (m->ctx[i]) vsc_ctx_destroy(m->ctx[i]);
    }
    delete m;
    return VSC_OK;
}

int vsc_multi_release_scratch(vsc_multi *m)
{
    if (!m) return VSC_ERR_INVALID;
    if (!m->device.empty()) {
        (void)hipSetDevice(m->device[0]);
        for (auto &b : m->gbuf) b.release();
        for (auto &b : m->gvotes) b.release();
        m->votes_out.release();
        if (m->merge_ctx) (void)vsc_ctx_release_scratch(m->merge_ctx);
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void)hipSetDevice(m->device[i]);
        for (int k = 0; k < 2; ++k) {
            m->xbuf[k][i].release();
            m->vbuf[k][i].release();
        }
        if (m->ctx[i]) (void)vsc_ctx_release_scratch(m->ctx[i]);
    }
    return VSC_OK;
}

int vsc_multi_size(const vsc_multi *m) { return m ? (int)m->ctx.size() : 0; }
vsc_ctx *vsc_multi_ctx(vsc_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }
vsc_ctx *vsc_multi_result_ctx(vsc_multi *m) { return m ? m->merge_ctx : nullptr; }
const char *vsc_multi_last_error(const vsc_multi *m) { return m ? m->err.c_str() : "null multi-device context"; }
int vsc_multi_uses_rccl(const vsc_multi *m) { return m && m->use_rccl; }

int vsc_multi_get_timing(const vsc_multi *m, vsc_multi_timing *out)
{
    if (!m || !out) return VSC_ERR_INVALID;
    *out = m->timing;
    return VSC_OK;
}

int vsc_multi_genome_load(vsc_multi *m, const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t n_words,
                          const vsc_contig *contigs, uint32_t n_contigs, vsc_multi_genome **out)
{
    return mguarded(m, [&]() -> int {
    if (!m || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    m->err.clear();
    if (!hi || !lo || !nmask || !contigs || n_words == 0 || n_contigs == 0)
        return mfail(m, VSC_ERR_INVALID, "vsc_multi_genome_load: null or empty argument");
    std::unique_ptr<vsc_multi_genome, int (*)(vsc_multi_genome *)> holder(new (std::nothrow) vsc_multi_genome(), vsc_multi_genome_free);
    vsc_multi_genome *g = holder.get();
    if (!g) return mfail(m, VSC_ERR_NOMEM, "vsc_multi_genome_load: out of host memory");
    g->multi = m;
    const unsigned n = (unsigned)m->ctx.size();
    g->shard.assign(n, nullptr);
    std::vector<int> rc(n, VSC_OK);
    on_all(n, [&](size_t r) {
        uint64_t b, e;
        shard_range(n_words, (unsigned)r, n, &b, &e);
        if (e <= b) return;
        const uint64_t halo_end = std::min(e + 1, n_words);  // a 22-base halo = one word
        rc[r] = vsc_genome_load(m->ctx[r], hi + b, lo + b, nmask + b, b, halo_end - b, e - b, contigs, n_contigs, &g->shard[r]);
    });
    for (unsigned r = 0; r < n; ++r)
        if (rc[r] != VSC_OK) {
            return mfail(m, rc[r], "shard " + std::to_string(r) + ": " + vsc_last_error(m->ctx[r]));
        }
    const int trc = genome_table_only(m->merge_ctx, contigs, n_contigs, &g->table);
    if (trc != VSC_OK) return mfail(m, trc, std::string("contig table on the first device: ") + vsc_last_error(m->merge_ctx));
    *out = holder.release();
    return VSC_OK;
    });
}

int vsc_multi_genome_free(vsc_multi_genome *g)
{
    if (!g) return VSC_OK;
    for (vsc_genome *s : g->shard)
        if (s) vsc_genome_free(s);
    if (g->table) vsc_genome_free(g->table);
    delete g;
    return VSC_OK;
}

int vsc_multi_genome_build_index(vsc_multi *m, vsc_multi_genome *g, const vsc_search_params *params)
{
    return mguarded(m, [&]() -> int {
    if (!m || !g || g->multi != m) return VSC_ERR_INVALID;
    m->err.clear();
    const size_t n = m->ctx.size();
    std::vector<int> rc(n, VSC_OK);
    on_all(n, [&](size_t r) {
        if (g->shard[r]) rc[r] = vsc_genome_build_index(m->ctx[r], g->shard[r], params);
    });
    for (size_t r = 0; r < n; ++r)
        if (rc[r] != VSC_OK) return mfail(m, rc[r], "shard " + std::to_string(r) + ": " + vsc_last_error(m->ctx[r]));
    return VSC_OK;
    });
}

}  // extern "C"

namespace {

using clk = std::chrono::steady_clock;
double ms_between(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

// on_batch(hits, first read, reads, votes on the first device or null) -> status; *keep = true: the callback took the hits over
typedef std::function<int(vsc_hits *, uint32_t, uint32_t, const uint16_t *, bool *)> BatchSink;

// The engine behind vsc_multi_search and vsc_multi_search_stream (see the head of this file).
int run_batches(vsc_multi *m, const vsc_multi_genome *g, const uint64_t *guides, uint32_t n_guides, const vsc_search_params *params,
                uint32_t batch, const vsc_multi_score *score, const BatchSink &sink)
{
    const size_t n = m->ctx.size();
    const uint32_t mode = score ? score->mode : VSC_MULTI_SCORE_NONE;
    const bool votes = mode == VSC_MULTI_SCORE_VOTES;
    if (batch == 0 || batch > n_guides) batch = std::max<uint32_t>(n_guides, 1);
    const uint32_t n_batches = std::max<uint32_t>(1, (n_guides + batch - 1) / batch);
    const uint32_t k_max = 2 * batch;  // keys of a batch: read << 1 | strand
    const auto t0 = clk::now();

    // what the shard threads hand to the exchange thread, per slot
    struct Slot {
        std::vector<uint64_t> count;       // records of shard r
        std::vector<uint32_t> key_counts;  // [r * K + k], K = 2 x the batch's reads
    } slot[2];
    for (auto &s : slot) {
        s.count.assign(n, 0);
        s.key_counts.assign(n * (size_t)k_max, 0);
    }
    std::mutex mu;
    std::condition_variable cv;
    std::vector<uint32_t> produced(n, 0);  // batches shard r has searched, scored and packed
    uint32_t transferred = 0;              // batches whose exchange slots have been read out
    // (written under `mu`, which is what the waits re-check it under; atomic because this thread's loops also look at it between
    // two waits, without the lock)
    std::atomic<int> failed{VSC_OK};
    std::string why;
    std::vector<double> search_ms(n, 0), score_ms(n, 0);
    std::vector<uint64_t> shard_hits(n, 0);
    std::vector<clk::time_point> finished(n, t0);
    auto fail_all = [&](int code, const std::string &text) {
        std::lock_guard<std::mutex> lk(mu);
        if (failed == VSC_OK) {
            failed = code;
            why = text;
        }
        cv.notify_all();
    };

    auto shard_body = [&](size_t r) {
        if (hipSetDevice(m->device[r]) != hipSuccess) return fail_all(VSC_ERR_DEVICE, "shard " + std::to_string(r) + ": hipSetDevice failed");
        for (uint32_t b = 0; b < n_batches; ++b) {
            {
                std::unique_lock<std::mutex> lk(mu);  // this batch's slot was batch b - 2's: wait until that one has been sent
                cv.wait(lk, [&] { return failed != VSC_OK || transferred + 2 > b; });
                if (failed != VSC_OK) return;
            }
            const uint32_t first = b * batch, cnt = std::min<uint32_t>(batch, n_guides - std::min(n_guides, first)), K = 2 * cnt;
            Slot &s = slot[b & 1];
            s.count[r] = 0;
            std::fill(s.key_counts.begin() + r * (size_t)K, s.key_counts.begin() + (r + 1) * (size_t)K, 0u);
            if (g->shard[r]) {
                vsc_ctx *ctx = m->ctx[r];
                // what happens to one batch's result on its shard: the caller's per-hit scores where the hits were found, then
                // the records (+ votes) packed for the exchange
                auto pack = [&](vsc_hits *part, bool rows_done) -> int {
                    int rc = VSC_OK;
                    vsc_timing t{};
                    (void)vsc_ctx_timing(ctx, &t);
                    search_ms[r] += t.total_ms;
                    shard_hits[r] += t.hits;
                    const uint64_t c = vsc_hits_count(part);
                    s.count[r] = c;
                    if (c && mode == VSC_MULTI_SCORE_ROWS && !rows_done)
                        rc = vsc_score_hits_packed(ctx, g->shard[r], part, guides + first, cnt, 0, c, nullptr, nullptr, nullptr);
                    // (an exchange buffer that does not fit beside the context's pooled scratch: the scratch goes back first)
                    auto ensure = [&](DeviceBuf &buf, size_t bytes) {
                        if (buf.ensure(bytes) == hipSuccess) return true;
                        (void)hipGetLastError();
                        (void)vsc_ctx_release_scratch(ctx);
                        return buf.ensure(bytes) == hipSuccess;
                    };
                    if (rc == VSC_OK && c && votes) {
                        if (!ensure(m->vbuf[b & 1][r], c * sizeof(uint16_t))) {
                            ctx->err = "vote buffer allocation failed";
                            rc = VSC_ERR_NOMEM;
                        } else {
                            rc = vsc_score_classify_hits(ctx, g->shard[r], part, guides + first, cnt, score->guide_activity + first, score->model,
                                                         0, c, m->vbuf[b & 1][r].p, nullptr, nullptr);
                        }
                    }
                    if (rc == VSC_OK && mode != VSC_MULTI_SCORE_NONE && c && !rows_done) {
                        (void)vsc_ctx_timing(ctx, &t);
                        score_ms[r] += t.score_ms;
                    }
                    if (rc == VSC_OK) {
                        if (!ensure(m->xbuf[b & 1][r], std::max<uint64_t>(c, 1) * VSC_XREC_BYTES)) {
                            ctx->err = "exchange buffer allocation failed";
                            rc = VSC_ERR_NOMEM;
                        } else {
                            rc = vsc_hits_pack_exchange(ctx, g->shard[r], part, cnt, m->xbuf[b & 1][r].p, 1, s.key_counts.data() + r * (size_t)K);
                        }
                    }
                    return rc;
                };
                int rc;
                if (mode == VSC_MULTI_SCORE_ROWS) {
                    // the feature rows are written on the way (vsc_search_stream_rows: one batch = the whole read range of this
                    // step), a consumer on the shard would read them inside the callback
                    struct Ctx {
                        decltype(pack) *fn;
                    } cb{&pack};
                    rc = vsc_search_stream_rows(ctx, g->shard[r], guides + first, cnt, params, cnt,
                                                [](void *u, vsc_hits *part, uint32_t, uint32_t, const void *) { return (*((Ctx *)u)->fn)(part, true); }, &cb);
                    if (rc == VSC_OK && cnt == 0) {  // (no reads: no batch, no callback)
                        s.count[r] = 0;
                    }
                } else {
                    vsc_hits *part = nullptr;
                    rc = vsc_search(ctx, g->shard[r], guides + first, cnt, params, &part);
                    if (rc == VSC_OK) {
                        rc = pack(part, false);
                        vsc_hits_free(part);  // the 8-byte records carry everything the merge needs
                    }
                }
                if (rc != VSC_OK) return fail_all(rc, "shard " + std::to_string(r) + ": " + vsc_last_error(ctx));
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                produced[r] = b + 1;
                finished[r] = clk::now();
            }
            cv.notify_all();
        }
    };
    auto shard_loop = [&](size_t r) {  // (no exception leaves a thread: the host allocations in there are strings and small vectors)
        try {
            shard_body(r);
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (failed == VSC_OK) failed = VSC_ERR_NOMEM;
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(n);
    // whatever ends this scope - also an exception on this thread, or a shard thread that cannot be started -: the shard
    // threads are told to stop (they wait on `failed`) before they are joined
    struct Joiner {
        std::vector<std::thread> &p;
        std::mutex &mu;
        std::condition_variable &cv;
        std::atomic<int> &failed;
        bool done = false;
        ~Joiner()
        {
            if (!done) {
                std::lock_guard<std::mutex> lk(mu);
                if (failed == VSC_OK) failed = VSC_ERR_NOMEM;
            }
            cv.notify_all();
            for (auto &t : p)
                if (t.joinable()) t.join();
        }
    } joiner{pool, mu, cv, failed};
    for (size_t r = 0; r < n; ++r) pool.emplace_back(shard_loop, r);

    // ---- the exchange thread: this one ---------------------------------------------------------------------------
    vsc_multi_timing mt{};
    auto hip_fail = [&](const char *what, hipError_t e) {
        fail_all(e == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
        return false;
    };
#define VSC_X(call)                                            \
    do {                                                       \
        const hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return hip_fail(#call, e_);      \
    } while (0)
    // Inside an RCCL group nothing may return early: an open group makes every later call on these communicators queue
    // forever.  The first failure is kept, the group is closed, and the communicators are given up (a half-issued
    // send / receive pairing cannot be repaired): later searches use device copies.
    auto give_up_rccl = [&](const char *where, ncclResult_t bad) {
        for (ncclComm_t &c : m->comm) {
            if (c && m->rccl.CommAbort) (void)m->rccl.CommAbort(c);
            c = nullptr;
        }
        m->comm.clear();
        m->use_rccl = false;
        fail_all(VSC_ERR_DEVICE, std::string(where) + ": " + m->rccl.GetErrorString(bad) + " (RCCL given up, later searches copy)");
        return false;
    };
    // shard r's records (+ votes) of the batch in `s` to the first device
    auto transfer = [&](size_t r, uint32_t b, const Slot &s) -> bool {
        const uint64_t c = s.count[r];
        if (!c) return true;
        const size_t rec_bytes = c * VSC_XREC_BYTES, vote_bytes = votes ? c * sizeof(uint16_t) : 0;
        VSC_X(hipSetDevice(m->device[0]));
        VSC_X(m->gbuf[r].ensure(rec_bytes));
        if (votes) VSC_X(m->gvotes[r].ensure(vote_bytes));
        const void *src = m->xbuf[b & 1][r].p, *vsrc = m->vbuf[b & 1][r].p;
        // (RCCL with ONE device - the hook rccl = 1 / 2 on a one-GPU box - sends to itself, so that the calls are exercised there)
        if (m->device[r] == m->device[0] && !(m->use_rccl && n == 1)) {
            VSC_X(hipMemcpyAsync(m->gbuf[r].p, src, rec_bytes, hipMemcpyDeviceToDevice, m->xstream[0]));
            if (votes) VSC_X(hipMemcpyAsync(m->gvotes[r].p, vsrc, vote_bytes, hipMemcpyDeviceToDevice, m->xstream[0]));
            if (r != 0) mt.exchanged_bytes += rec_bytes + vote_bytes;  // (another context on the same device: a rehearsal)
        } else if (m->use_rccl) {
            ncclResult_t bad = ncclSuccess;
            auto in_group = [&](ncclResult_t x) {
                if (bad == ncclSuccess && x != ncclSuccess) bad = x;
            };
            in_group(m->rccl.GroupStart());
            if (bad == ncclSuccess) {
                // (one thread drives both ends: the current device is set to the communicator's before each call, as
                // RCCL's single-process examples do)
                (void)hipSetDevice(m->device[0]);
                in_group(m->rccl.Recv(m->gbuf[r].p, rec_bytes, ncclUint8, (int)r, m->comm[0], m->xstream[0]));
                if (votes) in_group(m->rccl.Recv(m->gvotes[r].p, vote_bytes, ncclUint8, (int)r, m->comm[0], m->xstream[0]));
                (void)hipSetDevice(m->device[r]);
                in_group(m->rccl.Send(src, rec_bytes, ncclUint8, 0, m->comm[r], m->xstream[r]));
                if (votes) in_group(m->rccl.Send(vsrc, vote_bytes, ncclUint8, 0, m->comm[r], m->xstream[r]));
                in_group(m->rccl.GroupEnd());
                (void)hipSetDevice(m->device[0]);
            }
            if (bad != ncclSuccess) return give_up_rccl("send / receive of the hit records", bad);
            if (r != 0) mt.exchanged_bytes += rec_bytes + vote_bytes;
        } else {
            VSC_X(hipMemcpyPeerAsync(m->gbuf[r].p, m->device[0], src, m->device[r], rec_bytes, m->xstream[0]));
            if (votes) VSC_X(hipMemcpyPeerAsync(m->gvotes[r].p, m->device[0], vsrc, m->device[r], vote_bytes, m->xstream[0]));
            mt.exchanged_bytes += rec_bytes + vote_bytes;
        }
        return true;
    };
    auto finish_transfers = [&]() -> bool {  // everything issued for this batch has arrived / left
        VSC_X(hipSetDevice(m->device[0]));
        VSC_X(hipStreamSynchronize(m->xstream[0]));
        if (m->use_rccl)
            for (size_t r = 1; r < n; ++r) {
                VSC_X(hipSetDevice(m->device[r]));
                VSC_X(hipStreamSynchronize(m->xstream[r]));
            }
        return true;
    };
#undef VSC_X

    for (uint32_t b = 0; b < n_batches && failed == VSC_OK; ++b) {
        const uint32_t first = b * batch, cnt = std::min<uint32_t>(batch, n_guides - std::min(n_guides, first)), K = 2 * cnt;
        const Slot &s = slot[b & 1];
        std::vector<char> sent(n, 0);
        size_t n_sent = 0;
        bool ok = true;
        clk::time_point all_ready = clk::now();
        while (n_sent < n && ok) {
            std::vector<size_t> ready;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] {
                    if (failed != VSC_OK) return true;
                    for (size_t r = 0; r < n; ++r)
                        if (!sent[r] && produced[r] > b) return true;
                    return false;
                });
                if (failed != VSC_OK) break;
                for (size_t r = 0; r < n; ++r)
                    if (!sent[r] && produced[r] > b) ready.push_back(r);
            }
            all_ready = clk::now();
            for (size_t r : ready) {  // a shard's records leave the moment it is done - the others are still sorting
                ok = ok && transfer(r, b, s);
                sent[r] = 1;
                ++n_sent;
            }
        }
        if (failed != VSC_OK || !ok) break;
        if (!finish_transfers()) break;
        const auto t_arrived = clk::now();
        mt.exchange_ms += ms_between(all_ready, t_arrived);
        std::vector<uint32_t> kc(s.key_counts.begin(), s.key_counts.begin() + n * (size_t)K);  // (the slot is reused two batches on)
        {
            std::lock_guard<std::mutex> lk(mu);
            transferred = b + 1;
        }
        cv.notify_all();
        // ---- merge on the first device: shards partition the positions in ascending order ------------------------------
        std::vector<const void *> rec_ptr(n), vote_ptr(n);
        for (size_t r = 0; r < n; ++r) {
            rec_ptr[r] = m->gbuf[r].p;
            vote_ptr[r] = m->gvotes[r].p;
        }
        vsc_hits *merged = nullptr;
        const int mrc = merge_packed_shards(m->merge_ctx, g->table, rec_ptr.data(), votes ? vote_ptr.data() : nullptr, kc.data(), (uint32_t)n,
                                            2 * first, K, &merged, votes ? &m->votes_out : nullptr);
        if (mrc != VSC_OK) {
            fail_all(mrc, std::string("merge: ") + vsc_last_error(m->merge_ctx));
            break;
        }
        const auto t_merged = clk::now();
        mt.merge_ms += ms_between(t_arrived, t_merged);
        bool keep = false;
        const int crc = sink(merged, first, cnt, votes && vsc_hits_count(merged) ? (const uint16_t *)m->votes_out.p : nullptr, &keep);
        if (!keep) vsc_hits_free(merged);
        mt.callback_ms += ms_between(t_merged, clk::now());
        if (crc != VSC_OK) {
            fail_all(crc, "the batch callback stopped the stream");
            break;
        }
        mt.batches++;
    }
    joiner.done = true;  // (a stream that broke off has set `failed` itself)
    for (auto &t : pool) t.join();
    // (the exchange and landing buffers stay pooled, like every context's scratch: hipMalloc / hipFree of 13 GB cost hundreds
    // of milliseconds per search - vsc_multi_release_scratch gives them back)
    if (failed != VSC_OK) return mfail(m, failed, why);
    clk::time_point last = t0;
    for (size_t r = 0; r < n; ++r) {
        last = std::max(last, finished[r]);
        mt.search_ms_max = std::max(mt.search_ms_max, search_ms[r]);
        mt.score_ms_max = std::max(mt.score_ms_max, score_ms[r]);
        mt.hits += shard_hits[r];
    }
    mt.search_wall_ms = ms_between(t0, last);
    mt.total_ms = ms_between(t0, clk::now());
    mt.n_devices = (uint32_t)n;
    mt.used_rccl = m->use_rccl;
    m->timing = mt;
    return VSC_OK;
}

}  // namespace

extern "C" {

int vsc_multi_search(vsc_multi *m, const vsc_multi_genome *g, const uint64_t *guides, uint32_t n_guides,
                     const vsc_search_params *params, vsc_hits **out)
{
    return mguarded(m, [&]() -> int {
    if (!m || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    m->err.clear();
    if (!g || g->multi != m || !params || (n_guides && !guides)) return mfail(m, VSC_ERR_INVALID, "vsc_multi_search: null argument");
    // one batch = one exchange; a read set beyond one search pass is handled inside every shard's vsc_search
    return run_batches(m, g, guides, n_guides, params, 0, nullptr, [&](vsc_hits *h, uint32_t, uint32_t, const uint16_t *, bool *keep) {
        *out = h;
        *keep = true;
        return VSC_OK;
    });
    });
}

int vsc_multi_search_stream(vsc_multi *m, const vsc_multi_genome *g, const uint64_t *guides, uint32_t n_guides,
                            const vsc_search_params *params, uint32_t batch_reads, const vsc_multi_score *score,
                            vsc_multi_batch_fn on_batch, void *user)
{
    return mguarded(m, [&]() -> int {
    if (!m) return VSC_ERR_INVALID;
    m->err.clear();
    if (!g || g->multi != m || !params || (n_guides && !guides) || !on_batch)
        return mfail(m, VSC_ERR_INVALID, "vsc_multi_search_stream: null argument");
    if (score && score->mode > VSC_MULTI_SCORE_VOTES) return mfail(m, VSC_ERR_INVALID, "vsc_multi_search_stream: unknown scoring mode");
    if (score && score->mode == VSC_MULTI_SCORE_VOTES && (!score->model || (n_guides && !score->guide_activity)))
        return mfail(m, VSC_ERR_INVALID, "vsc_multi_search_stream: the votes need a forest and the reads' activities");
    if (batch_reads == 0 || batch_reads > (uint32_t)kMaxPassReads) batch_reads = kMaxPassReads;
    return run_batches(m, g, guides, n_guides, params, batch_reads, score,
                       [&](vsc_hits *h, uint32_t first, uint32_t cnt, const uint16_t *votes_dev, bool *) { return on_batch(user, h, first, cnt, votes_dev); });
    });
}

}  // extern "C"
