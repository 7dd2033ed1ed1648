// vcf_expand.hpp - alt-allele expansion (row R8): one sample column of a VCF -> the "SNP genome", a set
// of short windows holding every haplotype combination of nearby variants.  Host C++17.
//
// Semantics follow VARSCOT_pipeline/variant_processing/process_vcf.h:54-269 (record -> variants),
// overlap_sequences.h:35-240 (sweep over position-sorted variants) and write_fasta.h:30-470
// (haplotype combinations, window sequences, FASTA ids) as chained by vcf_loader.cpp:40-68.
// Where the reference has undefined behaviour the definitions of DESIGN.md section 8 are used
// (maxDeletion[-1] = 0; a record's variants share one pos; the out-of-bounds variants[1] write keeps
// the second-allele variant; the position sort is stable).
//
// Reference bases come from a caller-supplied fetch(chromosome, begin, end) with FAI clamping
// (write_fasta.h:245-271), so the tool can serve them from a FASTA held in memory or packed planes.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace vsc_vcf {

struct Variant {
    std::string ref, alt;
    uint32_t pos = 0, start = 0, end = 0;
    int type = 0;    // 0 substitution, 1 insertion, 2 deletion (process_vcf.h:196-207)
    int allele = 0;  // 0 first, 1 second, 2 both, -1 unphased (:191-194)
};

using Record = std::vector<Variant>;  // the 1-2 variants of one VCF line

// most unphased records one range may hold: 2^n windows are written for n of them (expand_range)
constexpr size_t kMaxUnphasedInRange = 24;

struct Chromosome {
    std::string name;
    std::vector<Record> records;  // in file order
};

inline std::string dna5(const std::string &s)
{
    std::string o(s);
    for (auto &c : o) {
        switch (c) {
        case 'A': case 'a': c = 'A'; break;
        case 'C': case 'c': c = 'C'; break;
        case 'G': case 'g': c = 'G'; break;
        case 'T': case 't': c = 'T'; break;
        default: c = 'N';
        }
    }
    return o;
}

inline std::vector<std::string> split(const std::string &s, char sep)
{
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = s.find(sep, b);
        out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}

// What `std::istringstream is(text); is >> int; is >> char >> int;` does for a GT field, without the stream
// (constructing one per line was the largest single cost of parsing a 5-million-record VCF): leading white
// space skipped, optional sign, decimal digits; no digits = failure, after which every later read fails too.
struct GtScanner {
    const char *p, *e;
    bool ok = true;
    explicit GtScanner(const std::string &s) : p(s.data()), e(s.data() + s.size()) {}
    void skip_space()
    {
        while (p < e && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
    }
    bool read_int(int &v)
    {
        if (!ok) return false;
        skip_space();
        const char *q = p;
        bool neg = false;
        if (q < e && (*q == '+' || *q == '-')) neg = *q++ == '-';
        if (q >= e || *q < '0' || *q > '9') return ok = false;
        long long x = 0;
        while (q < e && *q >= '0' && *q <= '9') {
            if (x < (1ll << 40)) x = x * 10 + (*q - '0');
            ++q;
        }
        p = q;
        if (x > 2147483647ll) {  // out of range: the stream stores the limit and sets failbit
            v = neg ? -2147483647 - 1 : 2147483647;
            return ok = false;
        }
        v = (int)(neg ? -x : x);
        return true;
    }
    bool read_char(char &c)
    {
        if (!ok) return false;
        skip_space();
        if (p >= e) return ok = false;
        c = *p++;
        return true;
    }
};

// The five columns of a VCF data line that matter (POS, REF, ALT, FORMAT, the sample's column) -> the record's
// variants (empty = the sample carries no alternative here).
inline Record parse_columns(const std::string &pos_col, const std::string &ref_col, const std::string &alt_col,
                            const std::string &format_col, const std::string &sample_col)
{
    Record out;
    Variant base;
    base.pos = (uint32_t)(std::strtoul(pos_col.c_str(), nullptr, 10) - 1);
    base.ref = dna5(ref_col);
    // index of GT among the FORMAT keys (0 when absent), then that field of the sample's column
    size_t gt_at = 0;
    {
        size_t i = 0, b = 0;
        for (;; ++i) {
            const size_t e = format_col.find(':', b);
            if (format_col.compare(b, (e == std::string::npos ? format_col.size() : e) - b, "GT") == 0) {
                gt_at = i;
                break;
            }
            if (e == std::string::npos) break;
            b = e + 1;
        }
    }
    std::string gt_text;
    {
        size_t i = 0, b = 0;
        for (;; ++i) {
            const size_t e = sample_col.find(':', b);
            if (i == gt_at) {
                gt_text = sample_col.substr(b, e == std::string::npos ? std::string::npos : e - b);
                break;
            }
            if (e == std::string::npos) break;
            b = e + 1;
        }
    }
    const auto alts = split(alt_col, ',');
    GtScanner is(gt_text);
    int first = -1, second = -1;
    char sep = 0;
    bool phased = true;
    if (!is.read_int(first) || first < 0 || (size_t)first > alts.size()) return out;
    if ((is.read_char(sep) && is.read_int(second)) && second >= 0 && (size_t)second <= alts.size()) {
        if (sep == '/') phased = false;
    } else {
        second = first;  // haploid call (process_vcf.h:104-108)
    }
    auto make = [&](int allele, const std::string &alt) {
        Variant v = base;
        v.allele = allele;
        v.alt = dna5(alt);
        out.push_back(v);
    };
    if (first == 0 && second == 0) return out;
    if (first > 0 && second > 0 && first != second) {
        const std::string &a1 = alts[first - 1], &a2 = alts[second - 1];
        if (a1 != "." && a2 != ".") {
            make(0, a1);
            make(1, a2);
        } else if (a1 != ".") {
            make(0, a1);
        } else if (a2 != ".") {
            make(1, a2);
        } else {
            return out;
        }
    } else {
        if (alts[0] == ".") return out;
        if (first == 0)
            make(1, alts[second - 1]);
        else if (second == 0)
            make(0, alts[first - 1]);
        else
            make(2, alts[first - 1]);
    }
    for (auto &v : out) {
        if (!phased && first != second) v.allele = -1;
        v.type = v.ref.size() > v.alt.size() ? 2 : (v.ref.size() == v.alt.size() ? 0 : 1);
    }
    return out;
}

// One VCF data line, split at tabs -> the record's variants.
inline Record parse_record(const std::vector<std::string> &f, unsigned sample)
{
    if (f.size() < 10 || sample >= f.size() - 9) throw std::out_of_range("ERROR: Sample index out of range.");
    return parse_columns(f[1], f[3], f[4], f[8], f[9 + sample]);
}

// Reads a VCF.  Chromosome order = ##contig header order, then order of first appearance.
inline std::vector<Chromosome> read_vcf(std::istream &in, unsigned sample)
{
    std::vector<Chromosome> chrs;
    std::unordered_map<std::string, size_t> at;
    auto chr_of = [&](const std::string &name) -> Chromosome & {
        auto it = at.find(name);
        if (it == at.end()) {
            it = at.emplace(name, chrs.size()).first;
            chrs.push_back({name, {}});
        }
        return chrs[it->second];
    };
    std::string line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (line.empty()) continue;
        if (line.compare(0, 2, "##") == 0) {
            if (line.compare(0, 10, "##contig=<") == 0) {
                size_t p = line.find("ID=");
                if (p != std::string::npos) {
                    size_t e = line.find_first_of(",>", p);
                    chr_of(line.substr(p + 3, e == std::string::npos ? std::string::npos : e - p - 3));
                }
            }
            continue;
        }
        if (line[0] == '#') continue;
        const auto f = split(line, '\t');
        if (f.size() < 2) continue;
        Chromosome &c = chr_of(f[0]);
        Record r = parse_record(f, sample);
        if (!r.empty()) c.records.push_back(std::move(r));
    }
    return chrs;
}

struct Range {
    uint32_t first, last;  // half-open range of position-sorted records
    uint32_t center;       // index (into the sorted order's underlying records) of the centre record
};

// The sweep of overlap_sequences.h:35-162 over one chromosome.  `order` = record indices sorted by
// position (stable).  Sets start / end on the centre records.
inline std::vector<Range> sweep(std::vector<Record> &recs, const std::vector<uint32_t> &order, uint32_t seq_len)
{
    const int n = (int)order.size();
    std::vector<uint32_t> max_del(n, 0);
    for (int i = 0; i < n; ++i)
        for (const auto &v : recs[order[i]])
            if (v.type == 2) max_del[i] = std::max<uint32_t>(max_del[i], (uint32_t)(v.ref.size() - v.alt.size()));
    auto md = [&](int k) -> uint32_t { return (k >= 0 && k < n) ? max_del[k] : 0u; };
    auto pos = [&](int k) -> uint32_t { return recs[order[k]][0].pos; };
    std::vector<Range> out;
    uint32_t r1 = 0, r2 = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t w_left, w_right;
        if (r2 > (uint32_t)i) {
            int right = (int)r2;
            w_right = seq_len + max_del[i];
            if (right < n)
                for (int d = i + 1; d <= right; ++d) w_right += max_del[d];
            while (right < n && (uint32_t)(pos(right) - pos(i)) < w_right) {
                w_right += max_del[right];
                ++right;
            }
            if ((uint32_t)right == r2) {  // nothing new to the right: only the previous window grows
                for (auto &v : recs[out.back().center]) v.end = pos(i) + w_right;
                continue;
            }
            r2 = (uint32_t)right;
            int left = i - 1;
            w_left = seq_len + md(left);
            while (left >= 0 && (uint32_t)(pos(i) - pos(left)) < w_left) {
                --left;
                w_left += md(left);
            }
            if ((uint32_t)(left + 1) == r1) {  // same left edge as the previous range: extend it
                for (auto &v : recs[out.back().center]) v.end = pos(i) + w_right;
                out.back().last = r2;
                continue;
            }
            r1 = (uint32_t)(left + 1);
        } else {
            w_right = seq_len + max_del[i];
            int right = i + 1;
            while (right < n && (uint32_t)(pos(right) - pos(i)) < w_right) {
                w_right += max_del[right];
                ++right;
            }
            r2 = (uint32_t)right;
            w_left = seq_len;
            r1 = (uint32_t)i;
        }
        out.push_back({r1, r2, order[i]});
        for (auto &v : recs[order[i]]) {
            v.start = v.pos - w_left + 1;  // unsigned wrap-around near a contig start, as in the reference
            v.end = v.pos + w_right;
        }
    }
    return out;
}

using Fetch = std::function<std::string(const std::string &chr, uint32_t begin, uint32_t end)>;
using Emit = std::function<void(const std::string &id, const std::string &seq)>;

// write_fasta.h:30-65
inline std::string fasta_id(const std::vector<Record> &recs, const std::vector<uint32_t> &order, const Range &rg,
                            const std::vector<int> &choice, const std::string &chr)
{
    std::string id = chr + "_" + std::to_string(recs[rg.center][0].start) + "_";
    if (std::all_of(choice.begin(), choice.end(), [](int c) { return c == -1; })) return id + "REF";
    id += "ALT";
    for (size_t i = 0; i < choice.size(); ++i)
        if (choice[i] != -1) {
            const Variant &v = recs[order[rg.first + i]][choice[i]];
            id += "_" + std::to_string(v.pos) + "_" + v.ref + "_" + v.alt;
        }
    return id;
}

// All windows of one range (write_fasta.h:88-229, 303-399).
inline void expand_range(const std::vector<Record> &recs, const std::vector<uint32_t> &order, const Range &rg,
                         const std::string &chr, const Fetch &fetch, const Emit &emit)
{
    const uint32_t i1 = rg.first, i2 = rg.last, size = i2 - i1;
    const Variant &center = recs[rg.center][0];
    const bool start_variant = center.start > recs[order[i1]][0].pos;
    const bool end_variant = center.end == recs[order[i2 - 1]][0].pos;
    // reference pieces: before the first variant (unless the window starts inside it), between
    // variants, after the last one (unless the window ends at it)
    std::vector<std::string> piece;  // alternating reference / allele slots
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
        uint32_t b, e;
        if (j == 0) {
            b = center.start;
            e = recs[order[i]][0].pos;
        } else {
            const Variant &prev = recs[order[i - 1]][0];
            b = prev.pos + (uint32_t)prev.ref.size();
            e = i == i2 ? center.end : recs[order[i]][0].pos;
        }
        piece[j] = fetch(chr, b, e);
    }
    // haplotype choices: phased records are fixed, unphased ones enumerated 0..0 -> 1..1
    std::vector<std::string> a1(size), a2(size);
    std::vector<int> c1(size, 0), c2(size, 0);
    std::vector<uint32_t> unphased;
    for (uint32_t k = 0; k < size; ++k) {
        const Record &r = recs[order[i1 + k]];
        if (r[0].allele == -1) {
            unphased.push_back(k);
        } else if (r.size() == 2) {
            a1[k] = r[0].alt, c1[k] = 0, a2[k] = r[1].alt, c2[k] = 1;
        } else if (r[0].allele == 0) {
            a1[k] = r[0].alt, c1[k] = 0, a2[k] = r[0].ref, c2[k] = -1;
        } else if (r[0].allele == 1) {
            a1[k] = r[0].ref, c1[k] = -1, a2[k] = r[0].alt, c2[k] = 0;
        } else {
            a1[k] = r[0].alt, a2[k] = r[0].alt, c1[k] = 0, c2[k] = 0;
        }
    }
    auto write = [&](const std::vector<std::string> &alleles, const std::vector<int> &choice) {
        std::string seq;
        for (uint32_t k = 0, slot = 1 - ref_start; k < size; ++k, slot += 2) piece[slot] = alleles[k];
        for (const auto &p : piece) seq += p;
        emit(fasta_id(recs, order, rg, choice, chr), seq);
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
            const int bit = (int)((mask >> (unphased.size() - 1 - u)) & 1);  // first unphased variant varies slowest
            const uint32_t k = unphased[u];
            const Record &r = recs[order[i1 + k]];
            if (r.size() == 2) {
                a1[k] = a2[k] = r[bit].alt;
                c1[k] = c2[k] = bit;
            } else if (bit == 0) {
                a1[k] = a2[k] = r[0].ref;
                c1[k] = c2[k] = -1;
            } else {
                a1[k] = a2[k] = r[0].alt;
                c1[k] = c2[k] = 0;
            }
        }
        both();
    }
}

// vcf_loader.cpp:40-68: every window of every chromosome, in the reference's output order.
inline void expand(std::vector<Chromosome> &chrs, uint32_t seq_len, const Fetch &fetch, const Emit &emit)
{
    for (auto &c : chrs) {
        std::vector<uint32_t> order(c.records.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(),
                         [&](uint32_t a, uint32_t b) { return c.records[a][0].pos < c.records[b][0].pos; });
        const auto ranges = sweep(c.records, order, seq_len);
        for (const auto &rg : ranges) expand_range(c.records, order, rg, c.name, fetch, emit);
    }
}

}  // namespace vsc_vcf
