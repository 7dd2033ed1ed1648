// stub_search.cpp - genome, search and result entry points for varscot_pipeline's sanitizer run (tools/multi_tsan/run.sh): a
// brute-force comparison of every read with every 23-base window of the (small) test genome on the HOST, so that the
// one-process driver's own code - records -> potential off-targets on all host threads, vsc_windows_build (the real
// csrc/vsc_windows.cpp), the shadow table, coordinate restoration, output text - runs under ASan / UBSan / TSan without a GPU.
// TEST INFRASTRUCTURE ONLY and not a restatement of anything: the acceptance rule is a plausible simplification (no N in the
// window, PAM GG / GA / the extra one, at most m differences over the 23 positions); nothing here is linked into the product.
#include <algorithm>
#include <string>
#include <vector>

#include "stub_host.h"
#include "varscot_hip.h"

struct vsc_genome {
    std::vector<uint32_t> hi, lo, nm;
    std::vector<vsc_contig> contigs;
};
struct vsc_hits {
    std::vector<vsc_hit> rec;
};
struct vsc_multi;
struct vsc_multi_genome;

extern "C" {
int vsc_genome_load(vsc_ctx *, const uint32_t *hi, const uint32_t *lo, const uint32_t *nm, uint64_t, uint64_t n_words, uint64_t, const vsc_contig *contigs,
                    uint32_t n_contigs, vsc_genome **out)
{
    vsc_genome *g = new vsc_genome();
    g->hi.assign(hi, hi + n_words);
    g->lo.assign(lo, lo + n_words);
    g->nm.assign(nm, nm + n_words);
    g->contigs.assign(contigs, contigs + n_contigs);
    *out = g;
    return VSC_OK;
}
int vsc_genome_free(vsc_genome *g)
{
    delete g;
    return VSC_OK;
}
int vsc_genome_index_load(vsc_ctx *ctx, vsc_genome *, const char *)
{
    ctx->err = "stub: no seed index files";
    return VSC_ERR_INVALID;
}

int vsc_search(vsc_ctx *, const vsc_genome *g, const uint64_t *guides, uint32_t n_guides, const vsc_search_params *p, vsc_hits **out)
{
    auto rc = [](char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; };
    std::vector<std::string> reads(n_guides, std::string(VSC_READ_LEN, 'N'));
    for (uint32_t i = 0; i < n_guides; ++i) {
        for (int j = 0; j < VSC_READ_LEN; ++j) reads[i][j] = "ACGT"[(guides[i] >> (2 * j)) & 3u];  // (vsc_pack_guide: base j in bits 2j, 2j + 1)
    }
    vsc_hits *res = new vsc_hits();
    for (uint32_t c = 0; c < g->contigs.size(); ++c) {
        const vsc_contig &ct = g->contigs[c];
        if (ct.length < VSC_READ_LEN) continue;
        std::string text(ct.length, 'N');
        vsc_unpack_bases(g->hi.data(), g->lo.data(), g->nm.data(), ct.offset, ct.length, &text[0]);
        for (uint32_t pos = 0; pos + VSC_READ_LEN <= ct.length; ++pos) {
            const char *w = &text[pos];
            if (std::find(w, w + VSC_READ_LEN, 'N') != w + VSC_READ_LEN) continue;
            for (uint32_t strand = 0; strand < 2; ++strand) {
                char site[VSC_READ_LEN];
                for (int j = 0; j < VSC_READ_LEN; ++j) site[j] = strand ? rc(w[VSC_READ_LEN - 1 - j]) : w[j];
                const bool pam = site[21] == 'G' ? (site[22] == 'G' || site[22] == 'A') : false;
                if (!pam && !(p->has_extra_pam && site[21] == p->extra_pam[0] && site[22] == p->extra_pam[1])) continue;
                for (uint32_t i = 0; i < n_guides; ++i) {
                    uint32_t mask = 0, nm = 0;
                    for (int j = 0; j < VSC_READ_LEN && nm <= p->max_mismatches; ++j)
                        if (site[j] != reads[i][j]) {
                            mask |= 1u << (strand ? VSC_READ_LEN - 1 - j : j);  // forward-window coordinates
                            ++nm;
                        }
                    if (nm > p->max_mismatches) continue;
                    res->rec.push_back(vsc_hit{i, c, pos, strand << 31 | nm << 23 | mask});
                }
            }
        }
    }
    std::sort(res->rec.begin(), res->rec.end(), [](const vsc_hit &a, const vsc_hit &b) {
        const uint32_t sa = a.info >> 31, sb = b.info >> 31;
        if (a.guide != b.guide) return a.guide < b.guide;
        if (sa != sb) return sa < sb;
        if (a.contig != b.contig) return a.contig < b.contig;
        return a.pos < b.pos;
    });
    *out = res;
    return VSC_OK;
}
uint64_t vsc_hits_count(const vsc_hits *h) { return h ? h->rec.size() : 0; }
int vsc_hits_data(vsc_hits *h, const vsc_hit **out)
{
    *out = h->rec.data();
    return VSC_OK;
}
int vsc_hits_free(vsc_hits *h)
{
    delete h;
    return VSC_OK;
}

// (the device list of -D is not part of this run)
int vsc_multi_create(const int *, int, vsc_multi **) { return VSC_ERR_NODEVICE; }
int vsc_multi_destroy(vsc_multi *) { return VSC_OK; }
int vsc_multi_genome_load(vsc_multi *, const uint32_t *, const uint32_t *, const uint32_t *, uint64_t, const vsc_contig *, uint32_t, vsc_multi_genome **) { return VSC_ERR_NODEVICE; }
int vsc_multi_genome_free(vsc_multi_genome *) { return VSC_OK; }
int vsc_multi_search(vsc_multi *, const vsc_multi_genome *, const uint64_t *, uint32_t, const vsc_search_params *, vsc_hits **) { return VSC_ERR_NODEVICE; }
const char *vsc_multi_last_error(const vsc_multi *) { return "stub: one device only"; }
}

namespace vsc {
bool host_timing_on() { return false; }  // (csrc/vsc_windows.cpp asks; defined in vsc_api.cpp in the library)
}

extern "C" vsc_ctx *vsc_multi_ctx(vsc_multi *, int) { return nullptr; }  // (bidir_mapping -D list: not part of these runs)
