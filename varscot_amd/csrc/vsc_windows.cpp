// vsc_windows.cpp - the alt-allele windows of one VCF sample ("SNP genome", row R8) built straight into
// packed planes: what `vcf_loader FILE.vcf SNP.fa GENOME.fa SAMPLE 23 T` followed by `bidir_index -G SNP.fa`
// produces (VARSCOT_pipeline/variant_processing/vcf_loader.cpp:40-68, write_fasta.h:245-399; VARSCOT:296-307),
// without the FASTA text in between: reference segments are copied bit-wise from the resident reference
// planes instead of per-segment FAI reads (write_fasta.h:245-271), alleles are packed as they are placed,
// and every stage runs on all host threads (the reference's THREADS argument).
//
// Host C++ only.  The expansion rules themselves - record -> variants, overlap sweep, haplotype combinations,
// ids - are the ones of tools/vcf_expand.hpp (shared with the vcf_loader drop-in, which keeps the FASTA route);
// this file adds the parallel driver and the bit-level sink.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "tools/vcf_expand.hpp"
#include "varscot_hip.h"

namespace vsc {
bool host_timing_on();  // vsc_api.cpp (vsc_debug_set_host_timing)
}

namespace {

using namespace vsc_vcf;

struct RefView {
    const uint32_t *hi, *lo, *nm;
};

// appends bases to a unit's local planes (bit 0 of word 0 = the unit's first position; everything starts as N)
struct BitWriter {
    std::vector<uint32_t> hi, lo, nm;
    uint64_t pos = 0;

    void reserve_bits(uint64_t upto)
    {
        const size_t need = (size_t)((upto + 31) / 32) + 1;
        if (hi.size() < need) {
            const size_t to = std::max(need, hi.size() * 2 + 64);
            hi.resize(to, 0u);
            lo.resize(to, 0u);
            nm.resize(to, 0xFFFFFFFFu);
        }
    }
    // n <= 32 bases given as plane bits (bit i = base i); N positions have m = 1 and h = l = 0
    void put(uint32_t h, uint32_t l, uint32_t m, unsigned n)
    {
        if (n == 0) return;
        reserve_bits(pos + n);
        const uint32_t valid = n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
        const uint64_t base = ~m & valid;  // positions that become real bases
        const size_t w = (size_t)(pos >> 5);
        const unsigned sh = (unsigned)(pos & 31);
        const uint64_t H = (uint64_t)(h & base) << sh, L = (uint64_t)(l & base) << sh, B = base << sh;
        hi[w] |= (uint32_t)H;
        lo[w] |= (uint32_t)L;
        nm[w] &= ~(uint32_t)B;
        if (sh + n > 32) {
            hi[w + 1] |= (uint32_t)(H >> 32);
            lo[w + 1] |= (uint32_t)(L >> 32);
            nm[w + 1] &= ~(uint32_t)(B >> 32);
        }
        pos += n;
    }
    void put_ref(const RefView &r, uint64_t src, uint64_t n)
    {
        while (n) {
            const unsigned take = (unsigned)std::min<uint64_t>(n, 32);
            const size_t w = (size_t)(src >> 5);
            const unsigned sh = (unsigned)(src & 31);
            auto window = [&](const uint32_t *p) {  // `take` bits from bit position src
                uint64_t v = p[w];
                if (sh + take > 32) v |= (uint64_t)p[w + 1] << 32;  // never reads past the word that holds the last base
                return (uint32_t)(v >> sh);
            };
            put(window(r.hi), window(r.lo), window(r.nm), take);
            src += take;
            n -= take;
        }
    }
    void put_text(const std::string &s)
    {
        size_t i = 0;
        while (i < s.size()) {
            uint32_t h = 0, l = 0, m = 0;
            const unsigned take = (unsigned)std::min<size_t>(32, s.size() - i);
            for (unsigned b = 0; b < take; ++b) {
                switch (s[i + b]) {
                case 'A': break;
                case 'C': l |= 1u << b; break;
                case 'G': h |= 1u << b; break;
                case 'T': h |= 1u << b; l |= 1u << b; break;
                default: m |= 1u << b;
                }
            }
            put(h, l, m, take);
            i += take;
        }
    }
    void separator()
    {
        reserve_bits(pos + 1);
        ++pos;  // stays N
    }
};

// The windows of a block of ranges: their lengths, ids and bases (local planes).
struct Unit {
    uint32_t chr = 0, first_range = 0, last_range = 0;
    std::vector<uint32_t> lens;
    std::string names;              // ids, each followed by '\n'
    BitWriter bits;
};

struct ChrWork {
    std::vector<uint32_t> order;
    std::vector<Range> ranges;
    int ref_contig = -1;  // index into the reference contig table (-1: the chromosome is not in the genome)
};

// Window assembly without strings: the pieces of expand_range (tools/vcf_expand.hpp) as reference
// ranges / allele texts, written through the BitWriter.  Must stay in step with expand_range - the tests
// compare the two routes bit for bit.
void expand_range_packed(const std::vector<Record> &recs, const std::vector<uint32_t> &order, const Range &rg,
                         const std::string &chr, const RefView &ref, const vsc_contig *contig, Unit &out)
{
    const uint32_t i1 = rg.first, i2 = rg.last, size = i2 - i1;
    const Variant &center = recs[rg.center][0];
    const bool start_variant = center.start > recs[order[i1]][0].pos;
    const bool end_variant = center.end == recs[order[i2 - 1]][0].pos;
    struct Piece {
        uint32_t b = 0, e = 0;          // reference slot: [b, e) of the chromosome (unclamped)
        const std::string *text = nullptr;  // allele slot
    };
    std::vector<Piece> piece;
    uint32_t ref_start, r_start, r_end;
    if (start_variant && end_variant) {
        piece.resize(2 * size - 1);
        ref_start = 1, r_start = i1 + 1, r_end = i2;
    } else if (start_variant) {
        piece.resize(2 * size);
        ref_start = 1, r_start = i1 + 1, r_end = i2 + 1;
    } else if (end_variant) {
        piece.resize(2 * size);
        ref_start = 0, r_start = i1, r_end = i2;
    } else {
        piece.resize(2 * size + 1);
        ref_start = 0, r_start = i1, r_end = i2 + 1;
    }
    for (uint32_t i = r_start, j = ref_start; i < r_end; ++i, j += 2) {
        if (j == 0) {
            piece[j].b = center.start;
            piece[j].e = recs[order[i]][0].pos;
        } else {
            const Variant &prev = recs[order[i - 1]][0];
            piece[j].b = prev.pos + (uint32_t)prev.ref.size();
            piece[j].e = i == i2 ? center.end : recs[order[i]][0].pos;
        }
    }
    std::vector<const std::string *> a1(size), a2(size);
    std::vector<int> c1(size, 0), c2(size, 0);
    std::vector<uint32_t> unphased;
    for (uint32_t k = 0; k < size; ++k) {
        const Record &r = recs[order[i1 + k]];
        if (r[0].allele == -1) {
            unphased.push_back(k);
        } else if (r.size() == 2) {
            a1[k] = &r[0].alt, c1[k] = 0, a2[k] = &r[1].alt, c2[k] = 1;
        } else if (r[0].allele == 0) {
            a1[k] = &r[0].alt, c1[k] = 0, a2[k] = &r[0].ref, c2[k] = -1;
        } else if (r[0].allele == 1) {
            a1[k] = &r[0].ref, c1[k] = -1, a2[k] = &r[0].alt, c2[k] = 0;
        } else {
            a1[k] = &r[0].alt, a2[k] = &r[0].alt, c1[k] = 0, c2[k] = 0;
        }
    }
    const uint64_t clen = contig ? contig->length : 0;
    auto write = [&](const std::vector<const std::string *> &alleles, const std::vector<int> &choice) {
        for (uint32_t k = 0, slot = 1 - ref_start; k < size; ++k, slot += 2) piece[slot].text = alleles[k];
        const uint64_t begin = out.bits.pos;
        for (uint32_t j = 0; j < piece.size(); ++j) {
            const bool is_ref = (j & 1u) == ref_start;  // reference slots sit at ref_start, ref_start + 2, ...
            if (!is_ref) {
                out.bits.put_text(*piece[j].text);
            } else {
                if (!contig) throw std::out_of_range("ERROR: Index out of range.");
                uint64_t b = std::min<uint64_t>(piece[j].b, clen), e = std::min<uint64_t>(piece[j].e, clen);  // write_fasta.h:255-260
                if (b > e) e = b;
                out.bits.put_ref(ref, contig->offset + b, e - b);
            }
        }
        out.lens.push_back((uint32_t)(out.bits.pos - begin));
        out.bits.separator();
        out.names += fasta_id(recs, order, rg, choice, chr);
        out.names += '\n';
    };
    auto both = [&]() {
        write(a1, c1);
        if (c1 != c2) write(a2, c2);
    };
    if (unphased.empty()) {
        both();
        return;
    }
    // 2^n windows for n unphased records in one range (write_fasta.h:155-229 enumerates them all - and keeps them in memory): a
    // VCF with dozens of unphased calls inside one window would fill the disk (or the memory) before anything is searched, and
    // 64 of them overflow the shift.  Such a range is refused.
    if (unphased.size() > kMaxUnphasedInRange)
        throw std::runtime_error("ERROR: " + std::to_string(unphased.size()) + " unphased variants within one window at " + chr + ":" +
                                 std::to_string(center.pos + 1) + " (2^" + std::to_string(unphased.size()) + " allele combinations; the limit is 2^" +
                                 std::to_string(kMaxUnphasedInRange) + ").");
    const uint64_t combos = 1ull << unphased.size();
    for (uint64_t mask = 0; mask < combos; ++mask) {
        for (size_t u = 0; u < unphased.size(); ++u) {
            const int bit = (int)((mask >> (unphased.size() - 1 - u)) & 1);
            const uint32_t k = unphased[u];
            const Record &r = recs[order[i1 + k]];
            if (r.size() == 2) {
                a1[k] = a2[k] = &r[bit].alt;
                c1[k] = c2[k] = bit;
            } else if (bit == 0) {
                a1[k] = a2[k] = &r[0].ref;
                c1[k] = c2[k] = -1;
            } else {
                a1[k] = a2[k] = &r[0].alt;
                c1[k] = c2[k] = 0;
            }
        }
        both();
    }
}

template <class F> void run_parallel(unsigned threads, size_t n_items, F &&body, std::string *error)
{
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::vector<std::string> errs(threads);
    auto worker = [&](unsigned tid) {
        try {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n_items || failed.load()) break;
                body(i);
            }
        } catch (const std::exception &e) {
            errs[tid] = e.what();
            failed.store(true);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto &t : pool) t.join();
    if (failed.load() && error)
        for (const auto &e : errs)
            if (!e.empty()) {
                *error = e;
                break;
            }
}

// One VCF sample column -> chromosomes with their records, parsed on all threads (line order preserved).
std::vector<Chromosome> read_vcf_parallel(const std::string &path, unsigned sample, unsigned threads)
{
    const bool timing = vsc::host_timing_on();
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vsc windows]   %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    std::string text;
    {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("ERROR: Could not open VCF file.");
        std::fseek(f, 0, SEEK_END);
        const long size = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        text.resize(size > 0 ? (size_t)size : 0);
        const size_t got = text.empty() ? 0 : std::fread(&text[0], 1, text.size(), f);
        std::fclose(f);
        text.resize(got);
    }
    lap("read file");
    // chunks that end on line boundaries
    const size_t n_chunks = std::max<size_t>(1, std::min<size_t>(threads * 8, text.size() / (1 << 20) + 1));
    std::vector<size_t> cut(n_chunks + 1, text.size());
    cut[0] = 0;
    for (size_t c = 1; c < n_chunks; ++c) {
        size_t p = text.size() / n_chunks * c;
        p = std::max(p, cut[c - 1]);
        const size_t nl = text.find('\n', p);
        cut[c] = nl == std::string::npos ? text.size() : nl + 1;
    }
    struct Event {
        std::string chr;  // a ##contig header (record empty) or a data line's chromosome
        Record record;
        bool header = false;
    };
    std::vector<std::vector<Event>> events(n_chunks);
    std::string error;
    run_parallel(threads, n_chunks, [&](size_t c) {
        auto &ev = events[c];
        size_t p = cut[c];
        const size_t end = cut[c + 1];
        ev.reserve((end - p) / 30 + 16);
        std::string col[5];  // POS, REF, ALT, FORMAT, the sample's column
        while (p < end) {
            size_t nl = text.find('\n', p);
            if (nl == std::string::npos || nl > end) nl = end;
            size_t le = nl;
            while (le > p && (text[le - 1] == '\r' || text[le - 1] == '\n')) --le;
            const char *line = text.data() + p;
            const size_t n = le - p;
            p = nl + 1;
            if (n == 0) continue;
            if (n >= 2 && line[0] == '#' && line[1] == '#') {  // as read_vcf (tools/vcf_expand.hpp)
                const std::string l(line, n);
                if (l.compare(0, 10, "##contig=<") == 0) {
                    const size_t q = l.find("ID=");
                    if (q != std::string::npos) {
                        const size_t e = l.find_first_of(",>", q);
                        ev.push_back({l.substr(q + 3, e == std::string::npos ? std::string::npos : e - q - 3), {}, true});
                    }
                }
                continue;
            }
            if (line[0] == '#') continue;
            // the columns parse_record uses, cut out without splitting the whole line
            const unsigned want[5] = {1, 3, 4, 8, 9 + sample};
            unsigned field = 0, got = 0;
            const char *fb = line, *const lend = line + n;
            Event e;
            for (;;) {
                const char *fe = (const char *)std::memchr(fb, '\t', (size_t)(lend - fb));
                if (!fe) fe = lend;
                if (field == 0) e.chr.assign(fb, fe);
                if (got < 5 && field == want[got]) col[got++].assign(fb, fe);
                ++field;
                if (fe == lend) break;
                fb = fe + 1;
            }
            if (field < 2) continue;  // read_vcf skips lines without a second column
            if (field < 10 || sample >= field - 9) throw std::out_of_range("ERROR: Sample index out of range.");
            e.record = parse_columns(col[0], col[1], col[2], col[3], col[4]);
            ev.push_back(std::move(e));  // an empty record still introduces its chromosome (read_vcf: chr_of before parse)
        }
    }, &error);
    if (!error.empty()) throw std::runtime_error(error);
    lap("parse lines");
    std::vector<Chromosome> chrs;
    std::unordered_map<std::string, size_t> at;
    size_t last = (size_t)-1;
    for (auto &ev : events)
        for (auto &e : ev) {
            if (last == (size_t)-1 || chrs[last].name != e.chr) {
                auto it = at.find(e.chr);
                if (it == at.end()) {
                    it = at.emplace(e.chr, chrs.size()).first;
                    chrs.push_back({e.chr, {}});
                }
                last = it->second;
            }
            if (!e.header && !e.record.empty()) {
                auto &recs = chrs[last].records;
                if (recs.size() == recs.capacity()) recs.reserve(std::max<size_t>(1 << 16, recs.size() * 2));
                recs.push_back(std::move(e.record));
            }
        }
    lap("merge");
    return chrs;
}

}  // namespace

// (raw arrays, not vectors: hundreds of MB that the worker threads fill completely - zero-filling them first,
// on one thread, cost more than building the windows)
struct vsc_windows {
    std::unique_ptr<uint32_t[]> hi, lo, nm;
    uint64_t n_words = 0;
    std::unique_ptr<vsc_contig[]> contigs;
    uint64_t n = 0;
    std::unique_ptr<char[]> names;           // ids, each followed by '\n'
    std::unique_ptr<uint64_t[]> name_off;    // [n + 1]
};

extern "C" {

int vsc_windows_build(const char *vcf_path, uint32_t sample, uint32_t seq_len, uint32_t threads, const uint32_t *hi,
                      const uint32_t *lo, const uint32_t *nmask, const vsc_contig *contigs, const char *const *contig_names,
                      uint32_t n_contigs, vsc_windows **out, char *err, size_t err_len)
{
    auto fail = [&](int code, const std::string &what) {
        if (err && err_len) std::snprintf(err, err_len, "%s", what.c_str());
        return code;
    };
    if (!out) return VSC_ERR_INVALID;
    *out = nullptr;
    if (!vcf_path || !hi || !lo || !nmask || !contigs || !contig_names || n_contigs == 0 || seq_len == 0)
        return fail(VSC_ERR_INVALID, "vsc_windows_build: null or empty argument");
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = std::min(threads, 256u);
    const bool timing = vsc::host_timing_on();
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[vsc windows] %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    try {
        std::vector<Chromosome> chrs = read_vcf_parallel(vcf_path, sample, threads);
        lap("parse");
        std::unordered_map<std::string, int> by_name;  // first word of the FASTA id (FAI rule)
        for (uint32_t c = 0; c < n_contigs; ++c) {
            std::string n = contig_names[c] ? contig_names[c] : "";
            by_name.emplace(n.substr(0, n.find_first_of(" \t")), (int)c);
        }
        std::vector<ChrWork> work(chrs.size());
        std::string error;
        // position order + overlap sweep, one chromosome per task
        run_parallel(threads, chrs.size(), [&](size_t ci) {
            auto &c = chrs[ci];
            auto &w = work[ci];
            w.order.resize(c.records.size());
            for (uint32_t i = 0; i < w.order.size(); ++i) w.order[i] = i;
            std::stable_sort(w.order.begin(), w.order.end(),
                             [&](uint32_t a, uint32_t b) { return c.records[a][0].pos < c.records[b][0].pos; });
            w.ranges = sweep(c.records, w.order, seq_len);
            auto it = by_name.find(c.name);
            w.ref_contig = it == by_name.end() ? -1 : it->second;
        }, &error);
        if (!error.empty()) return fail(VSC_ERR_INVALID, error);
        lap("sort + sweep");
        // units = blocks of ranges, in output order
        constexpr uint32_t kRangesPerUnit = 4096;
        std::vector<Unit> units;
        for (size_t ci = 0; ci < chrs.size(); ++ci)
            for (uint32_t r = 0; r < work[ci].ranges.size(); r += kRangesPerUnit) {
                Unit u;
                u.chr = (uint32_t)ci;
                u.first_range = r;
                u.last_range = (uint32_t)std::min<size_t>(work[ci].ranges.size(), (size_t)r + kRangesPerUnit);
                units.push_back(std::move(u));
            }
        const RefView ref{hi, lo, nmask};
        run_parallel(threads, units.size(), [&](size_t ui) {
            Unit &u = units[ui];
            const auto &c = chrs[u.chr];
            const auto &w = work[u.chr];
            const vsc_contig *contig = w.ref_contig >= 0 ? &contigs[w.ref_contig] : nullptr;
            for (uint32_t r = u.first_range; r < u.last_range; ++r)
                expand_range_packed(c.records, w.order, w.ranges[r], c.name, ref, contig, u);
        }, &error);
        if (!error.empty()) return fail(VSC_ERR_INVALID, error);
        lap("expand");
        // layout: the units' bit streams follow each other
        std::unique_ptr<vsc_windows> res(new vsc_windows());
        uint64_t total_bits = 0, n_windows = 0, name_bytes = 0;
        std::vector<uint64_t> unit_bit(units.size()), unit_win(units.size()), unit_name(units.size());
        for (size_t ui = 0; ui < units.size(); ++ui) {
            unit_bit[ui] = total_bits;
            unit_win[ui] = n_windows;
            unit_name[ui] = name_bytes;
            total_bits += units[ui].bits.pos;
            n_windows += units[ui].lens.size();
            name_bytes += units[ui].names.size();
        }
        if (total_bits >= (1ull << 32) - 8192) return fail(VSC_ERR_RANGE, "vsc_windows_build: the windows exceed the 32-bit position space");
        if (n_windows >= (1ull << 32)) return fail(VSC_ERR_RANGE, "vsc_windows_build: too many windows");
        uint64_t n_words = (total_bits + 31) / 32;
        if (n_words == 0) n_words = 1;
        res->n_words = n_words;
        res->n = n_windows;
        res->hi.reset(new uint32_t[n_words]);
        res->lo.reset(new uint32_t[n_words]);
        res->nm.reset(new uint32_t[n_words]);
        res->contigs.reset(new vsc_contig[std::max<uint64_t>(n_windows, 1)]);
        res->names.reset(new char[std::max<uint64_t>(name_bytes, 1)]);
        res->name_off.reset(new uint64_t[n_windows + 1]);
        res->name_off[n_windows] = name_bytes;
        uint32_t *H = res->hi.get(), *L = res->lo.get(), *M = res->nm.get();
        // the words two units share (and the padding behind the last base) start as "all N"; every other word
        // is written whole by the one unit that owns it
        for (size_t ui = 0; ui < units.size(); ++ui) {
            if (units[ui].bits.pos == 0) continue;
            for (const uint64_t w : {unit_bit[ui] >> 5, (unit_bit[ui] + units[ui].bits.pos - 1) >> 5}) H[w] = 0u, L[w] = 0u, M[w] = 0xFFFFFFFFu;
        }
        for (uint64_t w = total_bits >> 5; w < n_words; ++w) H[w] = 0u, L[w] = 0u, M[w] = 0xFFFFFFFFu;
        run_parallel(threads, units.size(), [&](size_t ui) {
            const Unit &u = units[ui];
            // contig table + names
            uint64_t bit = unit_bit[ui], nb = unit_name[ui];
            size_t np = 0;
            for (size_t k = 0; k < u.lens.size(); ++k) {
                vsc_contig &c = res->contigs[unit_win[ui] + k];
                c.offset = bit;
                c.length = u.lens[k];
                c.reserved = 0;
                bit += (uint64_t)u.lens[k] + 1;
                res->name_off[unit_win[ui] + k] = nb + np;
                np = u.names.find('\n', np) + 1;
            }
            if (!u.names.empty()) std::memcpy(res->names.get() + nb, u.names.data(), u.names.size());
            // bits: the unit's stream shifted to its place; the words it shares with its neighbours atomically
            const uint64_t n = u.bits.pos;
            if (n == 0) return;
            const uint64_t first = unit_bit[ui];
            const unsigned sh = (unsigned)(first & 31);
            const size_t w0 = (size_t)(first >> 5), w_last = (size_t)((first + n - 1) >> 5);
            const size_t src_words = (size_t)((n + 31) / 32);
            for (size_t w = w0; w <= w_last; ++w) {
                const size_t j = w - w0;  // destination word w takes source bits [32 j - sh, 32 j - sh + 32)
                auto take = [&](const std::vector<uint32_t> &p, uint32_t fill) {
                    const uint64_t lo_w = j == 0 ? (uint64_t)fill : (j - 1 < src_words ? p[j - 1] : fill);
                    const uint64_t hi_w = j < src_words ? p[j] : fill;
                    if (sh == 0) return (uint32_t)hi_w;
                    return (uint32_t)(((hi_w << 32 | lo_w) >> (32 - sh)) & 0xFFFFFFFFu);
                };
                // positions of this word that belong to the unit
                const uint64_t wb = (uint64_t)w * 32, lo_bit = std::max(wb, first), hi_bit = std::min(wb + 32, first + n);
                const uint32_t own = (uint32_t)((hi_bit - lo_bit == 32) ? 0xFFFFFFFFull : (((1ull << (hi_bit - lo_bit)) - 1ull) << (lo_bit - wb)));
                const uint32_t h = take(u.bits.hi, 0u) & own, l = take(u.bits.lo, 0u) & own, m = take(u.bits.nm, 0xFFFFFFFFu) | ~own;
                if (own == 0xFFFFFFFFu) {
                    H[w] = h;
                    L[w] = l;
                    M[w] = m;
                } else {
                    __atomic_fetch_or(&H[w], h, __ATOMIC_RELAXED);
                    __atomic_fetch_or(&L[w], l, __ATOMIC_RELAXED);
                    __atomic_fetch_and(&M[w], m, __ATOMIC_RELAXED);
                }
            }
        }, &error);
        if (!error.empty()) return fail(VSC_ERR_INVALID, error);
        lap("stitch");
        *out = res.release();
        return VSC_OK;
    } catch (const std::bad_alloc &) {
        return fail(VSC_ERR_NOMEM, "vsc_windows_build: out of host memory");
    } catch (const std::exception &e) {
        return fail(VSC_ERR_INVALID, e.what());
    }
}

uint32_t vsc_windows_count(const vsc_windows *w) { return w ? (uint32_t)w->n : 0; }
uint64_t vsc_windows_words(const vsc_windows *w) { return w ? w->n_words : 0; }
const uint32_t *vsc_windows_plane(const vsc_windows *w, int which)
{
    if (!w) return nullptr;
    return which == 0 ? w->hi.get() : (which == 1 ? w->lo.get() : (which == 2 ? w->nm.get() : nullptr));
}
const vsc_contig *vsc_windows_contigs(const vsc_windows *w) { return w ? w->contigs.get() : nullptr; }
const char *vsc_windows_name(const vsc_windows *w, uint32_t i, uint32_t *len)
{
    if (!w || i >= w->n) return nullptr;
    if (len) *len = (uint32_t)(w->name_off[i + 1] - w->name_off[i] - 1);  // without the separator
    return w->names.get() + w->name_off[i];
}
const uint64_t *vsc_windows_name_offsets(const vsc_windows *w) { return w ? w->name_off.get() : nullptr; }
void vsc_windows_free(vsc_windows *w) { delete w; }

}  // extern "C"
