// driver.cpp - runs csrc/vsc_multi.cpp (the real engine) over the host stand-in of the device layer (stub_device.cpp) under
// ThreadSanitizer: plain and streamed searches over 1..7 shards (repeated device ids = device copies, distinct ids with the
// rccl hook off = peer copies), batches of several sizes incl. a ragged last one and more batches than exchange slots, votes
// travelling with the records, a callback that stops the stream, a shard whose search fails in the middle, an empty read set.
// Every merged batch is compared with the result computed here without threads.  TEST INFRASTRUCTURE ONLY (see run.sh).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "stub_device.h"
#include "varscot_hip_debug.h"
#include "vsc_objects.h"

namespace {
int failures = 0;
#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            ++failures;                           \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);    \
            std::fprintf(stderr, "\n");           \
        }                                         \
    } while (0)

struct Expect {
    std::vector<vsc_hit> hits;
    std::vector<uint16_t> votes;
};

// what one batch must look like after the merge: keys ascending, inside a key the shards in order
Expect expected(const std::vector<uint64_t> &codes, uint32_t first, uint32_t cnt, const std::vector<uint64_t> &shard_first_word)
{
    Expect e;
    for (uint32_t i = 0; i < cnt; ++i)
        for (uint32_t s = 0; s < 2; ++s)
            for (uint64_t fw : shard_first_word) {
                const uint64_t code = codes[first + i];
                const uint32_t c = stub::fake_count(code, s, fw);
                for (uint32_t j = 0; j < c; ++j) {
                    vsc_hit r{};
                    r.guide = first + i;
                    r.pos = (uint32_t)(fw * 32 + 7 * j + (uint32_t)(stub::mix(code + s) % 5));
                    r.info = s << 31;
                    e.hits.push_back(r);
                    e.votes.push_back(stub::fake_vote(code, s, r.pos));
                }
            }
    return e;
}

bool same(const vsc_hits *h, const Expect &e)
{
    if (h->n != e.hits.size()) return false;
    for (size_t i = 0; i < e.hits.size(); ++i)
        if (h->host[i].guide != e.hits[i].guide || h->host[i].pos != e.hits[i].pos || h->host[i].info != e.hits[i].info) return false;
    return true;
}

struct StreamState {
    const std::vector<uint64_t> *codes;
    const std::vector<uint64_t> *fw;
    bool votes;
    int stop_at;  // batch index at which the callback returns an error (-1: never)
    int seen = 0;
    uint32_t next_first = 0;
};

int on_batch(void *user, vsc_hits *batch, uint32_t first, uint32_t cnt, const uint16_t *votes_dev)
{
    StreamState &st = *(StreamState *)user;
    CHECK(first == st.next_first, "batch starts at read %u, expected %u", first, st.next_first);
    st.next_first = first + cnt;
    const Expect e = expected(*st.codes, first, cnt, *st.fw);
    CHECK(same(batch, e), "batch %d (reads %u..%u): %llu records, expected %zu (or different ones)", st.seen, first, first + cnt,
          (unsigned long long)batch->n, e.hits.size());
    if (st.votes && !e.hits.empty()) {
        CHECK(votes_dev != nullptr, "no votes with batch %d", st.seen);
        if (votes_dev)
            for (size_t i = 0; i < e.votes.size(); ++i)
                if (votes_dev[i] != e.votes[i]) {
                    CHECK(false, "vote %zu of batch %d", i, st.seen);
                    break;
                }
    }
    return st.seen++ == st.stop_at ? VSC_ERR_INVALID : VSC_OK;
}
}  // namespace

int main()
{
    const uint64_t n_words = 64 * 23 + 17;  // 24 tiles, the last one ragged
    std::vector<uint32_t> plane(n_words, 0);
    vsc_contig contig{0, (uint32_t)(n_words * 32), 0};
    vsc_search_params params{};
    params.max_mismatches = 4;
    std::vector<uint64_t> codes(173);
    for (size_t i = 0; i < codes.size(); ++i) codes[i] = stub::mix(i + 1) >> 18;
    vsc_rf_model model{};
    std::vector<double> activity(codes.size(), 0.5);
    int runs = 0;
    for (int n = 1; n <= 7; ++n)
        for (int distinct = 0; distinct < 2; ++distinct) {
            std::vector<int> ids(n, 0);
            if (distinct)
                for (int i = 0; i < n; ++i) ids[i] = i;
            vsc_multi_debug_params dbg{};
            dbg.rccl = 0;  // device copies / peer copies: RCCL itself is not what this run is about
            vsc_multi *m = nullptr;
            stub::ctx_serial = 0;
            CHECK(vsc_multi_create_debug(ids.data(), n, &dbg, &m) == VSC_OK && m, "create");
            if (!m) continue;
            vsc_multi_genome *g = nullptr;
            CHECK(vsc_multi_genome_load(m, plane.data(), plane.data(), plane.data(), n_words, &contig, 1, &g) == VSC_OK && g, "genome load");
            CHECK(vsc_multi_genome_build_index(m, g, &params) == VSC_OK, "index");
            std::vector<uint64_t> fw;
            const uint64_t tiles = (n_words + 63) / 64;
            for (int r = 0; r < n; ++r) {
                const uint64_t b = std::min<uint64_t>(tiles * r / n * 64, n_words), e = std::min<uint64_t>(tiles * (r + 1) / n * 64, n_words);
                if (e > b) fw.push_back(b);
            }
            // one batch
            vsc_hits *all = nullptr;
            CHECK(vsc_multi_search(m, g, codes.data(), (uint32_t)codes.size(), &params, &all) == VSC_OK && all, "search: %s", vsc_multi_last_error(m));
            if (all) {
                CHECK(same(all, expected(codes, 0, (uint32_t)codes.size(), fw)), "vsc_multi_search over %d shards", n);
                vsc_hits_free(all);
            }
            // streams: batch sizes that give 1, 2, 3, many batches (a ragged last one), every scoring mode
            for (uint32_t batch : {173u, 100u, 64u, 7u, 1u})
                for (uint32_t mode : {(uint32_t)VSC_MULTI_SCORE_NONE, (uint32_t)VSC_MULTI_SCORE_ROWS, (uint32_t)VSC_MULTI_SCORE_VOTES}) {
                    if (batch == 1u && (n > 3 || mode == VSC_MULTI_SCORE_ROWS)) continue;  // (173 batches: kept to a few combinations)
                    vsc_multi_score sc{};
                    sc.mode = mode;
                    sc.guide_activity = activity.data();
                    sc.model = &model;
                    StreamState st{&codes, &fw, mode == VSC_MULTI_SCORE_VOTES, -1};
                    const int rc = vsc_multi_search_stream(m, g, codes.data(), (uint32_t)codes.size(), &params, batch, &sc, on_batch, &st);
                    CHECK(rc == VSC_OK, "stream (n %d, batch %u, mode %u): %s", n, batch, mode, vsc_multi_last_error(m));
                    CHECK(st.seen == (int)((codes.size() + batch - 1) / batch) && st.next_first == codes.size(), "stream saw %d batches", st.seen);
                    vsc_multi_timing t{};
                    CHECK(vsc_multi_get_timing(m, &t) == VSC_OK && t.batches == (uint32_t)st.seen && t.n_devices == (uint32_t)n, "timing");
                    ++runs;
                }
            // the callback stops the stream at its third batch: an error, nobody left waiting, the next call works
            {
                StreamState st{&codes, &fw, false, 2};
                const int rc = vsc_multi_search_stream(m, g, codes.data(), (uint32_t)codes.size(), &params, 16, nullptr, on_batch, &st);
                CHECK(rc == VSC_ERR_INVALID && st.seen == 3, "stopped stream: rc %d after %d batches", rc, st.seen);
            }
            // a shard fails in the middle (its fourth batch)
            {
                stub::fail_shard = n - 1;
                stub::fail_code = codes[3 * 16];
                StreamState st{&codes, &fw, false, -1};
                const int rc = vsc_multi_search_stream(m, g, codes.data(), (uint32_t)codes.size(), &params, 16, nullptr, on_batch, &st);
                CHECK(rc == VSC_ERR_DEVICE && st.seen <= 3, "failing shard: rc %d after %d batches (%s)", rc, st.seen, vsc_multi_last_error(m));
                stub::fail_shard = -1;
                stub::fail_code = ~0ull;
            }
            // no reads; then an ordinary search again on the same object
            {
                vsc_hits *none = nullptr;
                CHECK(vsc_multi_search(m, g, codes.data(), 0, &params, &none) == VSC_OK && none && none->n == 0, "empty read set");
                if (none) vsc_hits_free(none);
                vsc_hits *again = nullptr;
                CHECK(vsc_multi_search(m, g, codes.data(), 50, &params, &again) == VSC_OK && again && same(again, expected(codes, 0, 50, fw)), "search after the failures");
                if (again) vsc_hits_free(again);
            }
            CHECK(vsc_multi_release_scratch(m) == VSC_OK, "release scratch");
            vsc_multi_genome_free(g);
            vsc_multi_destroy(m);
        }
    std::printf("multi_tsan: %d streamed runs over 1..7 shards, %llu merges, %llu queued copies performed, %d failures\n", runs,
                (unsigned long long)stub::merges.load(), (unsigned long long)stub::copies.load(), failures);
    return failures ? 1 : 0;
}
