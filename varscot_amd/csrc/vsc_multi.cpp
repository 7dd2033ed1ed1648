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
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
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
        for (const char *name : {only ? only : "librccl.so.1", only ? only : "librccl.so"}) {
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
    std::vector<hipStream_t> xstream;   // exchange streams, one per context (RCCL's streams)
    std::vector<uint64_t *> d_count;    // per context: [1 + n] 64-bit words - own count, then the all-gathered counts
    bool use_rccl = false;
    Rccl rccl;
    std::vector<ncclComm_t> comm;
    DeviceBuf gather;                   // on device[0]: the exchange records of all shards in shard order
    std::vector<DeviceBuf> xbuf;        // per context: its shard's 8-byte exchange records
    std::string err;
    vsc_multi_timing timing{};
};

struct vsc_multi_genome {
    vsc_multi *multi = nullptr;
    std::vector<vsc_genome *> shard;    // null: the shard owns no words of this (small) genome
    vsc_genome *table0 = nullptr;       // the contig table on the first device when shard[0] is null (for the merge)
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

template <class F> void on_all(size_t n, F &&body)
{
    std::vector<std::thread> pool;
    for (size_t i = 1; i < n; ++i) pool.emplace_back([&, i] { body(i); });
    if (n) body(0);
    for (auto &t : pool) t.join();
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
    vsc_multi *m = new (std::nothrow) vsc_multi();
    if (!m) return VSC_ERR_NOMEM;
    m->device.assign(device_ids, device_ids + n);
    m->ctx.assign(n, nullptr);
    m->xstream.assign(n, nullptr);
    m->d_count.assign(n, nullptr);
    m->xbuf.assign(n, DeviceBuf{});
    for (int i = 0; i < n; ++i) {
        const int rc = vsc_ctx_create(device_ids[i], &m->ctx[i]);
        if (rc != VSC_OK) {
            vsc_multi_destroy(m);
            return rc;
        }
        if (hipSetDevice(device_ids[i]) != hipSuccess || hipStreamCreate(&m->xstream[i]) != hipSuccess ||
            hipMalloc((void **)&m->d_count[i], (size_t)(1 + n) * sizeof(uint64_t)) != hipSuccess) {
            vsc_multi_destroy(m);
            return VSC_ERR_DEVICE;
        }
    }
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
            vsc_multi_destroy(m);
            return VSC_ERR_DEVICE;
        }
        m->use_rccl = ok;  // not forced and unavailable: peer copies carry the records instead
        if (!ok) m->err = why;  // (vsc_multi_last_error says why the copies are in use)
    } else if (forced) {
        std::fprintf(stderr, "vsc_multi_create: RCCL was asked for but needs distinct devices\n");
        vsc_multi_destroy(m);
        return VSC_ERR_INVALID;
    }
    *out = m;
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
        m->gather.release();
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void)hipSetDevice(m->device[i]);
        if (m->d_count[i]) (void)hipFree(m->d_count[i]);
        if (i < m->xbuf.size()) m->xbuf[i].release();
        if (m->xstream[i]) (void)hipStreamDestroy(m->xstream[i]);
        if (m->ctx[i]) vsc_ctx_destroy(m->ctx[i]);
    }
    delete m;
    return VSC_OK;
}

int vsc_multi_size(const vsc_multi *m) { return m ? (int)m->ctx.size() : 0; }
vsc_ctx *vsc_multi_ctx(vsc_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }
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
    vsc_multi_genome *g = new (std::nothrow) vsc_multi_genome();
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
            const std::string why = vsc_last_error(m->ctx[r]);
            vsc_multi_genome_free(g);
            return mfail(m, rc[r], "shard " + std::to_string(r) + ": " + why);
        }
    if (!g->shard[0]) {
        const int trc = genome_table_only(m->ctx[0], contigs, n_contigs, &g->table0);
        if (trc != VSC_OK) {
            const std::string why = vsc_last_error(m->ctx[0]);
            vsc_multi_genome_free(g);
            return mfail(m, trc, "contig table on the first device: " + why);
        }
    }
    *out = g;
    return VSC_OK;
    });
}

int vsc_multi_genome_free(vsc_multi_genome *g)
{
    if (!g) return VSC_OK;
    for (vsc_genome *s : g->shard)
        if (s) vsc_genome_free(s);
    if (g->table0) vsc_genome_free(g->table0);
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

int vsc_multi_search(vsc_multi *m, const vsc_multi_genome *g, const uint64_t *guides, uint32_t n_guides,
                     const vsc_search_params *params, vsc_hits **out)
{
    return mguarded(m, [&]() -> int {
    if (!m || !out) return VSC_ERR_INVALID;
    *out = nullptr;
    m->err.clear();
    if (!g || g->multi != m || !params || (n_guides && !guides)) return mfail(m, VSC_ERR_INVALID, "vsc_multi_search: null argument");
    const size_t n = m->ctx.size();
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    // ---- every device searches all reads on its shard and packs its records for the exchange ---------------------
    const uint32_t K = 2 * n_guides;
    std::vector<uint64_t> count(n, 0), off(n + 1, 0);
    std::vector<uint32_t> key_counts((size_t)n * K, 0);
    std::vector<int> rc(n, VSC_OK);
    std::vector<vsc_timing> st(n);
    on_all(n, [&](size_t r) {
        if (!g->shard[r]) return;
        vsc_hits *part = nullptr;
        rc[r] = vsc_search(m->ctx[r], g->shard[r], guides, n_guides, params, &part);
        if (rc[r] != VSC_OK) return;
        (void)vsc_ctx_timing(m->ctx[r], &st[r]);
        count[r] = vsc_hits_count(part);
        if (hipSetDevice(m->device[r]) != hipSuccess || m->xbuf[r].ensure(std::max<uint64_t>(count[r], 1) * VSC_XREC_BYTES) != hipSuccess) {
            m->ctx[r]->err = "exchange buffer allocation failed";
            rc[r] = VSC_ERR_NOMEM;
        } else {
            rc[r] = vsc_hits_pack_exchange(m->ctx[r], g->shard[r], part, n_guides, m->xbuf[r].p, 1, key_counts.data() + r * K);
        }
        vsc_hits_free(part);  // the 8-byte records carry everything the merge needs
    });
    for (size_t r = 0; r < n; ++r)
        if (rc[r] != VSC_OK) return mfail(m, rc[r], "shard " + std::to_string(r) + ": " + vsc_last_error(m->ctx[r]));
    const auto t1 = clock::now();
    vsc_multi_timing mt{};
    for (size_t r = 0; r < n; ++r) {
        off[r + 1] = off[r] + count[r];
        if (g->shard[r]) {
            mt.search_ms_max = std::max(mt.search_ms_max, st[r].total_ms);
            mt.hits += st[r].hits;
        }
    }
    const uint64_t total = off[n];
#define VSC_M(call)                                                                                                  \
    do {                                                                                                             \
        const hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                        \
            return mfail(m, e_ == hipErrorOutOfMemory ? VSC_ERR_NOMEM : VSC_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define VSC_N(call)                                                                                       \
    do {                                                                                                  \
        const ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) return mfail(m, VSC_ERR_DEVICE, std::string(#call) + ": " + m->rccl.GetErrorString(r_)); \
    } while (0)
    // ---- the one exchange: all records to the first device ------------------------------------------------
    VSC_M(hipSetDevice(m->device[0]));
    VSC_M(m->gather.ensure(std::max<uint64_t>(total, 1) * VSC_XREC_BYTES));
    char *const dst = (char *)m->gather.p;
    if (m->use_rccl) {
        // hit counts by all-gather (one 64-bit word per rank), checked against what this process already knows;
        // then one grouped send / receive per shard
        for (size_t r = 0; r < n; ++r) {
            VSC_M(hipSetDevice(m->device[r]));
            VSC_M(hipMemcpyAsync(m->d_count[r], &count[r], sizeof(uint64_t), hipMemcpyHostToDevice, m->xstream[r]));
        }
        // Inside a group nothing may return early: an open group makes every later RCCL call on these communicators
        // queue forever.  The first failure is kept, the group is closed, and the communicators are given up
        // (a half-issued send / receive pairing cannot be repaired): later searches use device copies.
        ncclResult_t bad = ncclSuccess;
        auto in_group = [&](ncclResult_t r) {
            if (bad == ncclSuccess && r != ncclSuccess) bad = r;
        };
        auto give_up_rccl = [&](const char *where) {
            for (ncclComm_t &c : m->comm) {
                if (c && m->rccl.CommAbort) (void)m->rccl.CommAbort(c);
                c = nullptr;
            }
            m->comm.clear();
            m->use_rccl = false;
            return mfail(m, VSC_ERR_DEVICE, std::string(where) + ": " + m->rccl.GetErrorString(bad) + " (RCCL given up, later searches copy)");
        };
        VSC_N(m->rccl.GroupStart());
        for (size_t r = 0; r < n && bad == ncclSuccess; ++r)
            in_group(m->rccl.AllGather(m->d_count[r], m->d_count[r] + 1, 1, ncclUint64, m->comm[r], m->xstream[r]));
        in_group(m->rccl.GroupEnd());
        if (bad != ncclSuccess) return give_up_rccl("all-gather of the hit counts");
        std::vector<uint64_t> seen(n, 0);
        VSC_M(hipSetDevice(m->device[0]));
        VSC_M(hipMemcpyAsync(seen.data(), m->d_count[0] + 1, n * sizeof(uint64_t), hipMemcpyDeviceToHost, m->xstream[0]));
        VSC_M(hipStreamSynchronize(m->xstream[0]));
        if (seen != count) return mfail(m, VSC_ERR_DEVICE, "vsc_multi_search: the all-gathered hit counts differ from the shards' counts");
        VSC_N(m->rccl.GroupStart());
        for (size_t r = 0; r < n && bad == ncclSuccess; ++r) {
            if (!count[r]) continue;
            in_group(m->rccl.Recv(dst + off[r] * VSC_XREC_BYTES, count[r] * VSC_XREC_BYTES, ncclUint8, (int)r, m->comm[0], m->xstream[0]));
            in_group(m->rccl.Send(m->xbuf[r].p, count[r] * VSC_XREC_BYTES, ncclUint8, 0, m->comm[r], m->xstream[r]));
        }
        in_group(m->rccl.GroupEnd());
        if (bad != ncclSuccess) return give_up_rccl("send / receive of the hit records");
        for (size_t r = 0; r < n; ++r) {
            VSC_M(hipSetDevice(m->device[r]));
            VSC_M(hipStreamSynchronize(m->xstream[r]));
        }
    } else {
        VSC_M(hipSetDevice(m->device[0]));
        for (size_t r = 0; r < n; ++r) {
            if (!count[r]) continue;
            if (m->device[r] == m->device[0])
                VSC_M(hipMemcpyAsync(dst + off[r] * VSC_XREC_BYTES, m->xbuf[r].p, count[r] * VSC_XREC_BYTES, hipMemcpyDeviceToDevice, m->xstream[0]));
            else
                VSC_M(hipMemcpyPeerAsync(dst + off[r] * VSC_XREC_BYTES, m->device[0], m->xbuf[r].p, m->device[r], count[r] * VSC_XREC_BYTES,
                                         m->xstream[0]));
        }
        VSC_M(hipStreamSynchronize(m->xstream[0]));
    }
    const auto t2 = clock::now();
    // ---- merge on the first device: shards partition the positions in ascending order ------------------------
    const vsc_genome *table = g->shard[0] ? g->shard[0] : g->table0;
    const int mrc = vsc_hits_merge_packed(m->ctx[0], table, dst, 1, key_counts.data(), (uint32_t)n, 0, K, out);
    // (the gathered records are not kept beside the merged result)
    if (total * VSC_XREC_BYTES > (64u << 20)) {
        (void)hipSetDevice(m->device[0]);
        m->gather.release();
    }
    if (mrc != VSC_OK) return mfail(m, mrc, std::string("merge: ") + vsc_last_error(m->ctx[0]));
    const auto t3 = clock::now();
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    mt.search_wall_ms = ms(t0, t1);
    mt.exchange_ms = ms(t1, t2);
    mt.merge_ms = ms(t2, t3);
    mt.total_ms = ms(t0, t3);
    mt.exchanged_bytes = (total - count[0]) * VSC_XREC_BYTES + (uint64_t)(n - 1) * K * sizeof(uint32_t);
    mt.n_devices = (uint32_t)n;
    mt.used_rccl = m->use_rccl;
    m->timing = mt;
    return VSC_OK;
#undef VSC_M
#undef VSC_N
    });
}

}  // extern "C"
