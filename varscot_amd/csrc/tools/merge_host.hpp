// merge_host.hpp - host side of the result mergers: SAM records -> potential off-targets, on-target /
// variant-window filters, TSV + feature-matrix text.  Semantics follow
// VARSCOT_pipeline/variant_processing/filter_output_bam.h:40-496 and merge_output_bam.h:46-720.
// Scores come from the GPU library (vsc_score_pairs); nothing is scored on the host.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <cstdio>
#include <functional>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "varscot_hip.h"
#include "vsc_host.hpp"

namespace vsc_merge {

struct OffTarget {  // PotentialOffTarget, filter_output_bam.h:28-37
    std::string chr, target, snp_type, sequence;
    std::vector<int> mm;  // mismatch positions, {-1} for a perfect match
    uint32_t pos = 0;
    char strand = '+';
};

inline bool same(const OffTarget &a, const OffTarget &b)  // comp(), :40-49
{
    return a.target == b.target && a.chr == b.chr && a.pos == b.pos && a.strand == b.strand && a.sequence == b.sequence &&
           a.mm == b.mm && a.snp_type == b.snp_type;
}

inline void revcomp_in_place(std::string &s)
{
    std::reverse(s.begin(), s.end());
    for (auto &c : s) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
}

// Sequences of a FASTA addressable by full id and by first word (FAI rule).  The bases come from the packed genome that
// stands for the FASTA when there is one (<prefix>.vsc of `bidir_index`, found by vsc_host::open_packed_for: a few mapped
// pages per region) - the part the FAI index plays in the reference (extract_fasta_ontargets.h:33-76,
// filter_output_bam.h:399) - and from the parsed FASTA text otherwise; same answers either way.
struct Genome {
    std::vector<vsc_host::FastaRecord> recs;  // text backend
    vsc_host::PackedView packed;              // packed backend
    bool from_packed = false;
    std::unordered_map<std::string, size_t> by_name;
    explicit Genome(const std::string &path, const char *packed_env = "VARSCOT_PACKED_GENOME")
    {
        from_packed = vsc_host::open_packed_for(path, packed_env, packed);
        if (!from_packed) recs = vsc_host::read_fasta(path);
        by_name.reserve(2 * size());
        for (size_t i = 0; i < size(); ++i) {
            const std::string &name = id(i);
            by_name.emplace(name, i);
            by_name.emplace(name.substr(0, name.find_first_of(" \t")), i);
        }
    }
    size_t size() const { return from_packed ? packed.names.size() : recs.size(); }
    const std::string &id(size_t i) const { return from_packed ? packed.names[i] : recs[i].id; }
    uint64_t length(size_t i) const { return from_packed ? packed.contigs[i].length : recs[i].seq.size(); }
    // extractSequenceFromIndex(..., flanking = false), extract_fasta_ontargets.h:33-76
    std::string region(const std::string &chr, uint32_t start, uint32_t end, char strand) const
    {
        auto it = by_name.find(chr);
        if (it == by_name.end()) throw std::out_of_range("ERROR: Index out of range.");
        return region_at(it->second, start, end, strand);
    }
    std::string region_at(size_t contig, uint32_t start, uint32_t end, char strand) const
    {
        const uint64_t len = length(contig);
        uint64_t b = std::min<uint64_t>(start, len), e = std::min<uint64_t>(end, len);
        if (b > e) e = b;
        std::string out;
        if (from_packed) {
            out = packed.bases((uint32_t)contig, b, e - b);
        } else {
            out = recs[contig].seq.substr(b, e - b);
            for (auto &c : out) {
                switch (c) {
                case 'A': case 'a': c = 'A'; break;
                case 'C': case 'c': c = 'C'; break;
                case 'G': case 'g': c = 'G'; break;
                case 'T': case 't': c = 'T'; break;
                default: c = 'N';
                }
            }
        }
        if (strand == '-') revcomp_in_place(out);
        return out;
    }
};

// getMismatchPositions, filter_output_bam.h:330-349: while (is >> num >> base) pos += num + 1
inline std::vector<int> md_positions(const std::string &md)
{
    std::vector<int> out;
    std::istringstream is(md);
    unsigned num = 0, pos = 0;
    char base = 0;
    while (is >> num >> base) {
        pos += num + 1;
        out.push_back((int)pos - 1);
    }
    if (out.empty()) out.push_back(-1);
    return out;
}

// readBamFile, filter_output_bam.h:362-418 (header-less SAM text as bidir_mapping writes it)
inline std::vector<OffTarget> read_sam(const std::string &path, const Genome &genome)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("ERROR: Could not open BAM file.");
    std::vector<OffTarget> out;
    std::string line;
    // A record that cannot be parsed ends the reading, as SeqAn's exception does in the reference (caught at :413-417: the
    // message goes to stdout, the records read so far are kept and the tool carries on with them).
    auto number = [](const std::string &s, unsigned long &v) {
        char *end = nullptr;
        v = std::strtoul(s.c_str(), &end, 10);
        return !s.empty() && end && *end == '\0';
    };
    for (size_t line_no = 1; std::getline(in, line); ++line_no) {
        if (line.empty() || line[0] == '@') continue;
        std::vector<std::string> f;
        size_t b = 0;
        for (;;) {
            size_t e = line.find('\t', b);
            f.push_back(line.substr(b, e == std::string::npos ? std::string::npos : e - b));
            if (e == std::string::npos) break;
            b = e + 1;
        }
        unsigned long flag = 0, pos1 = 0;
        if (f.size() < 11 || !number(f[1], flag) || !number(f[3], pos1) || pos1 == 0) {
            std::cout << "Malformed alignment record in line " << line_no << " of " << path << ": reading stops here." << std::endl;
            break;
        }
        OffTarget p;
        p.target = f[0];
        p.chr = f[2];
        p.pos = (uint32_t)(pos1 - 1);
        p.strand = (flag & 16u) ? '-' : '+';
        p.sequence = genome.region(p.chr, p.pos, p.pos + 23, p.strand);
        p.snp_type = "REF";
        std::string md;
        for (size_t i = 11; i < f.size(); ++i)
            if (f[i].compare(0, 5, "MD:Z:") == 0) md = f[i].substr(5);
        p.mm = md_positions(md);
        out.push_back(std::move(p));
    }
    return out;
}

// readOntargets, filter_output_bam.h:462-496 (BED6)
inline void read_ontargets(const std::string &path, const Genome &genome, std::map<std::string, OffTarget> &on,
                           std::map<std::string, unsigned> &count)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("ERROR: Could not open BED file.");
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line);
        std::string chr, name, score, strand;
        unsigned long start = 0, end = 0;
        if (!(is >> chr >> start >> end >> name >> score >> strand)) continue;
        OffTarget p;
        p.target = name;
        p.chr = chr;
        p.pos = (uint32_t)start;
        p.strand = strand.empty() ? '+' : strand[0];
        p.sequence = genome.region(chr, p.pos, p.pos + 23, p.strand);
        p.mm = {-1};
        p.snp_type = "REF";
        on.emplace(name, p);
        count.emplace(name, 0u);
    }
}

// readTuscanResult, feature_matrix.h:206-230
inline std::map<std::string, double> read_tuscan(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("ERROR: Could not open on-target activity file.");
    std::map<std::string, double> out;
    std::string line, target, sequence;
    double score;
    while (std::getline(in, line)) {
        std::istringstream is(line);
        if (is >> target >> sequence >> score) out.emplace(target, score);
    }
    return out;
}

inline int c_atoi(const std::string &s) { return (int)std::strtol(s.c_str(), nullptr, 10); }

inline std::vector<std::string> split_id(const std::string &id)
{
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = id.find('_', b);
        out.push_back(id.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}

// Variant windows per chromosome for the shadow filter (filterRefAlignment, :70-124): a reference hit
// is dropped if it lies fully inside any window of its chromosome.  The reference scans all windows
// per hit; here windows are sorted by start with a running maximum of their ends.
struct WindowIndex {
    struct Chr {
        std::vector<int64_t> start, max_end;
    };
    std::unordered_map<std::string, Chr> chrs;
    // the windows: id(i) = chr_start_..., length(i) = bases of the FASTA record (:443)
    template <class Id, class Len> WindowIndex(size_t n, Id &&id, Len &&length)
    {
        std::unordered_map<std::string, std::vector<std::pair<int64_t, int64_t>>> tmp;
        for (size_t i = 0; i < n; ++i) {
            // chr = the id up to its first '_', start = the number behind it (split on '_', :292,441)
            size_t len = 0;
            const char *name = id(i, &len);
            const char *u1 = (const char *)std::memchr(name, '_', len);
            if (!u1) continue;
            const char *rest = u1 + 1, *const end = name + len;
            const char *u2 = (const char *)std::memchr(rest, '_', (size_t)(end - rest));
            const int s = c_atoi(std::string(rest, u2 ? u2 : end));
            // `pos >= atoi(start)` (:102) compares unsigned with int: a negative start (a window whose
            // start wrapped around the contig start) is huge as unsigned and never <= pos
            if (s < 0) continue;
            tmp[std::string(name, u1)].push_back({(int64_t)s, (int64_t)s + (int64_t)length(i)});
        }
        for (auto &kv : tmp) {
            std::sort(kv.second.begin(), kv.second.end());
            Chr c;
            int64_t m = INT64_MIN;
            for (auto &w : kv.second) {
                m = std::max(m, w.second);
                c.start.push_back(w.first);
                c.max_end.push_back(m);
            }
            chrs.emplace(kv.first, std::move(c));
        }
    }
    explicit WindowIndex(const Genome &snp)
        : WindowIndex(snp.size(), [&](size_t i, size_t *len) { *len = snp.id(i).size(); return snp.id(i).data(); },
                      [&](size_t i) { return snp.length(i); })
    {
    }
    bool shadows(const std::string &chr, uint32_t pos, unsigned seq_len) const
    {
        auto it = chrs.find(chr);
        if (it == chrs.end()) return false;
        const auto &c = it->second;
        const size_t n = std::upper_bound(c.start.begin(), c.start.end(), (int64_t)pos) - c.start.begin();
        return n > 0 && (int64_t)pos + seq_len <= c.max_end[n - 1];
    }
};

// getSnpType, filter_output_bam.h:189-263
inline void snp_type(std::string &type, const std::vector<std::string> &id, uint32_t &pos, unsigned seq_len)
{
    std::string tag = "VAR_" + id[0] + "_";
    bool any = false, start_found = false;
    int count = 0;
    for (size_t i = 3; i + 2 < id.size(); i += 3) {
        const int p = c_atoi(id[i]);
        const size_t lr = id[i + 1].size(), la = id[i + 2].size();
        auto inside = [&](long q) { return (long)pos <= q && (long)pos + (long)seq_len > q; };
        if (lr == la) {
            if (inside(p)) {
                tag += id[i] + ",";
                any = start_found = true;
            }
        } else if (lr < la) {
            if (inside((long)p + 1) || inside((long)p + (long)la - 1)) {
                tag += id[i] + ",";
                any = start_found = true;
            } else if (!start_found) {
                count -= (int)(la - lr);
            }
        } else {
            if (inside((long)p + 1) || inside((long)p + (long)lr - 1)) {
                tag += id[i] + ",";
                any = start_found = true;
            } else if (!start_found) {
                count += (int)(lr - la);
            }
        }
    }
    pos += (uint32_t)count;
    if (any) type = tag.substr(0, tag.size() - 1);
}

inline std::string mm_columns(const OffTarget &p, bool merged)
{
    std::string s;
    if (p.mm.size() == 1 && p.mm[0] == -1) return merged ? "0\t\t" : "0\t";
    s = std::to_string(p.mm.size()) + "\t";
    for (size_t j = 0; j + 1 < p.mm.size(); ++j) s += std::to_string(p.mm[j]) + ",";
    s += std::to_string(p.mm.back());
    if (merged) s += "\t";
    return s;
}

inline uint32_t mm_mask(const OffTarget &p)
{
    uint32_t m = 0;
    for (int q : p.mm)
        if (q >= 0 && q < 32) m |= 1u << q;
    return m;
}

inline std::string fmt_double(double v)  // operator<<(ostream, double) with default precision
{
    std::ostringstream os;
    os << v;
    return os.str();
}

// getFeatureNames, feature_matrix.h:140-204
inline std::vector<std::string> feature_names()
{
    static const char *types[12] = {"AtoC", "AtoG", "AtoT", "CtoA", "CtoG", "CtoT", "GtoA", "GtoC", "GtoT", "TtoA", "TtoC", "TtoG"};
    static const char *letters[4] = {"A", "C", "G", "T"};
    std::vector<std::string> n(443);
    n[0] = "totalMismatches";
    for (int i = 1; i < 22; ++i) n[i] = "mismatchPos" + std::to_string(i);
    for (int i = 0; i < 12; ++i) n[22 + i] = types[i];
    n[34] = "transitionNumber";
    n[35] = "transversionNumber";
    for (int i = 1; i < 21; ++i)
        for (int j = 0; j < 4; ++j) n[36 + (i - 1) * 4 + j] = std::string(letters[j]) + std::to_string(i);
    n[116] = "PAMA", n[117] = "PAMC", n[118] = "PAMG", n[119] = "PAMT";
    for (int i = 1; i < 20; ++i)
        for (int j = 0; j < 16; ++j) n[120 + (i - 1) * 16 + j] = std::string(letters[j / 4]) + letters[j % 4] + std::to_string(i);
    for (int j = 0; j < 16; ++j) n[424 + j] = std::string(letters[j / 4]) + letters[j % 4];
    n[440] = "adjacentMismatches";
    n[441] = "seedMismatches";
    n[442] = "ontargetActivity";
    return n;
}

// What write_outputs may do beyond the two mergers' own output (all off by default = exactly the mergers' files):
struct OutputOptions {
    vsc_ctx *ctx = nullptr;  // score on this context (the caller's) instead of one created on `device` for the call
    // the random forest instead of "." in the Score column - what `classification_pipeline OUT FEATURES TRUE|FALSE` does to
    // the merger's files afterwards (classification/classificationPipeline.R:21-49): header "Name" instead of "Targetsite",
    // Score = probability of class "1" (prob) or the class
    const vsc_rf_model *forest = nullptr;
    bool prob = false;
    // rows in the order of the name column, byte-wise - what the driver's `LC_ALL=C sort -t TAB -k4,4` leaves (VARSCOT:355-357);
    // the feature matrix keeps the merge order
    bool sort_by_name = false;
};

// Scores rows on the GPU and writes TSV (+ feature matrix) text.  `merged` selects the 10-column
// layout of mergeResults (Variants column) over the 9-column one of processRefOnly.
inline void write_outputs(const std::string &out_path, const std::string *feature_path, bool merged,
                          const std::vector<const OffTarget *> &rows, const std::map<std::string, OffTarget> &on,
                          std::map<std::string, unsigned> &count, const std::map<std::string, double> &activity, int device,
                          const OutputOptions &opt = OutputOptions())
{
    std::ofstream out(out_path);
    std::ofstream fout;
    if (feature_path) fout.open(*feature_path);
    if (!out.is_open() || (feature_path && !fout.is_open())) throw std::runtime_error("ERROR: Could not open output file.");
    if (opt.forest && !feature_path) throw std::runtime_error("ERROR: the classifier needs the feature matrix.");
    out << (opt.forest ? "#Chr\tStart\tEnd\tName\tScore\tStrand\tSequence\tMismatch_Number\tMismatch_Positions"
                       : "#Chr\tStart\tEnd\tTargetsite\tScore\tStrand\tSequence\tMismatch_Number\tMismatch_Positions")
        << (merged ? "\tVariants\n" : "\n");
    if (feature_path) {
        const auto names = feature_names();
        for (size_t i = 0; i + 1 < names.size(); ++i) fout << names[i] << "\t";
        fout << names.back() << "\n";
    }
    const size_t n = rows.size();
    std::vector<double> mit(n);
    std::vector<uint8_t> feat(feature_path ? n * VSC_N_FEATURES : 0);
    std::unordered_map<std::string, std::string> activity_text;
    if (feature_path)
        for (size_t i = 0; i < n; ++i)
            if (!activity_text.count(rows[i]->target)) activity_text[rows[i]->target] = fmt_double(activity.at(rows[i]->target));
    std::vector<double> forest_prob;
    std::vector<uint8_t> forest_cls;
    if (n) {
        std::vector<uint64_t> on_codes(n), off_codes(n);
        std::vector<uint32_t> masks(n);
        for (size_t i = 0; i < n; ++i) {
            const OffTarget &p = *rows[i];
            const OffTarget &t = on.at(p.target);
            if (feature_path && (t.sequence.size() != VSC_READ_LEN || p.sequence.size() != VSC_READ_LEN))
                throw std::runtime_error("ERROR: a sequence is not 23 nt long.");
            std::string a = t.sequence, b = p.sequence;
            a.resize(VSC_READ_LEN, 'A');
            b.resize(VSC_READ_LEN, 'A');
            on_codes[i] = vsc_pack_guide(a.c_str());
            off_codes[i] = vsc_pack_guide(b.c_str());
            masks[i] = mm_mask(p);
        }
        vsc_ctx *ctx = opt.ctx;
        int st = VSC_OK;
        if (!ctx) st = vsc_ctx_create(device, &ctx);
        if (st != VSC_OK) throw std::runtime_error(st == VSC_ERR_NODEVICE ? "ERROR: no HIP device available (there is no CPU fallback)." : "ERROR: could not create the device context.");
        st = vsc_score_pairs(ctx, on_codes.data(), off_codes.data(), masks.data(), n, feature_path ? nullptr : mit.data(), nullptr,
                             feature_path ? feat.data() : nullptr);
        if (st == VSC_OK && opt.forest) {
            // the activity the classifier sees is the one the feature matrix carries: six significant digits
            std::vector<double> act(n);
            std::unordered_map<std::string, double> rounded;
            for (const auto &kv : activity_text) rounded[kv.first] = std::strtod(kv.second.c_str(), nullptr);
            for (size_t i = 0; i < n; ++i) act[i] = rounded.at(rows[i]->target);
            forest_prob.resize(n);
            forest_cls.resize(n);
            std::vector<uint8_t> tie(n);
            st = vsc_rf_predict(ctx, opt.forest, feat.data(), act.data(), n, forest_prob.data(), forest_cls.data(), tie.data());
        }
        const std::string err = st == VSC_OK ? "" : vsc_last_error(ctx);
        if (!opt.ctx) vsc_ctx_destroy(ctx);
        if (st != VSC_OK) throw std::runtime_error("ERROR: " + err);
    }
    // Text: the names carry running numbers per target (sequential), the rows themselves are formatted in blocks
    // by all host threads and written in order - 443 numbers per feature row through a stream, one `<<` each,
    // took 20 s for 2.6 M rows.
    std::vector<uint32_t> number(n);
    for (size_t i = 0; i < n; ++i) number[i] = ++count.at(rows[i]->target);
    char small[256][4];  // "0" .. "255"
    uint8_t small_len[256];
    for (int v = 0; v < 256; ++v) small_len[v] = (uint8_t)std::snprintf(small[v], sizeof small[v], "%d", v);
    auto score_text = [&](size_t i) -> std::string {
        if (opt.forest) {
            if (!opt.prob) return forest_cls[i] ? "1" : "0";  // an exact 500/500 vote (R: random) is reported as "0"
            char buf[64];
            std::snprintf(buf, sizeof buf, "%.15g", forest_prob[i]);
            return buf;
        }
        return feature_path ? std::string(".") : fmt_double(mit[i]);
    };
    auto tsv_row = [&](size_t i, const std::string &name, std::string &tsv) {
        const OffTarget &p = *rows[i];
        tsv += p.chr;
        tsv += '\t';
        tsv += std::to_string(p.pos);
        tsv += '\t';
        tsv += std::to_string(p.pos + 23);
        tsv += '\t';
        tsv += name;
        tsv += '\t';
        tsv += score_text(i);
        tsv += '\t';
        tsv += p.strand;
        tsv += '\t';
        tsv += p.sequence;
        tsv += '\t';
        tsv += mm_columns(p, merged);
        if (merged) tsv += p.snp_type;
        tsv += '\n';
    };
    // sorted output: every row's text is kept, with where its name starts and ends (the sort key)
    struct Placed { uint32_t block; uint32_t begin, name_b, name_e, end; };
    std::vector<Placed> placed(opt.sort_by_name ? n : 0);
    auto format_block = [&](size_t b, size_t e, std::string &tsv, std::string &fm, uint32_t block_id) {
        tsv.clear();
        fm.clear();
        if (feature_path) fm.reserve((e - b) * (2 * VSC_N_FEATURES + 64));
        for (size_t i = b; i < e; ++i) {
            const OffTarget &p = *rows[i];
            const std::string name = p.target + "_" + std::to_string(number[i]);
            const size_t at = tsv.size();
            tsv_row(i, name, tsv);
            if (opt.sort_by_name) {
                size_t nb = at;
                for (int tabs = 0; tabs < 3; ++nb)
                    if (tsv[nb] == '\t') ++tabs;
                placed[i] = Placed{block_id, (uint32_t)at, (uint32_t)nb, (uint32_t)(nb + name.size()), (uint32_t)tsv.size()};
            }
            if (feature_path) {
                fm += name;
                fm += '\t';
                const uint8_t *f = feat.data() + i * VSC_N_FEATURES;
                for (int k = 0; k < VSC_N_FEATURES; ++k) {
                    fm.append(small[f[k]], small_len[f[k]]);
                    fm += '\t';
                }
                fm += activity_text.at(p.target);
                fm += '\n';
            }
        }
    };
    const unsigned n_threads = vsc_host::host_threads();
    const size_t block = 16384;
    std::vector<std::string> kept;  // sorted output: the TSV text of every block
    if (opt.sort_by_name) kept.resize((n + block - 1) / block);
    // two sets of buffers: while one round of blocks is written (in order, by one thread), the next is formatted
    std::vector<std::string> tsv[2], fm[2];
    for (int k = 0; k < 2; ++k) {
        tsv[k].resize(n_threads);
        fm[k].resize(n_threads);
    }
    std::thread writer;
    struct JoinWriter {  // (an exception on the way out must not meet a running thread)
        std::thread &t;
        ~JoinWriter()
        {
            if (t.joinable()) t.join();
        }
    } join_writer{writer};
    int round = 0;
    for (size_t base = 0; base < n; base += block * n_threads, round ^= 1) {
        std::vector<std::thread> workers;
        unsigned used = 0;
        for (unsigned t = 0; t < n_threads && base + t * block < n; ++t, ++used) {
            const size_t b = base + t * block, e = std::min(n, b + block);
            std::string &dst = opt.sort_by_name ? kept[b / block] : tsv[round][t];
            workers.emplace_back(format_block, b, e, std::ref(dst), std::ref(fm[round][t]), (uint32_t)(b / block));
        }
        for (auto &w : workers) w.join();
        if (writer.joinable()) writer.join();  // (the round before has left its buffers)
        writer = std::thread([&, used, round] {
            for (unsigned t = 0; t < used; ++t) {
                if (!opt.sort_by_name) out.write(tsv[round][t].data(), (std::streamsize)tsv[round][t].size());
                if (feature_path) fout.write(fm[round][t].data(), (std::streamsize)fm[round][t].size());
            }
        });
    }
    if (writer.joinable()) writer.join();
    if (opt.sort_by_name) {
        // `sort -k4,4` under LC_ALL=C: the name field byte-wise, ties (there are none: names are unique) by the whole line
        std::vector<uint32_t> order(n);
        for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        auto key_less = [&](uint32_t x, uint32_t y) {
            const Placed &a = placed[x], &b = placed[y];
            const std::string &sa = kept[a.block], &sb = kept[b.block];
            const size_t la = a.name_e - a.name_b, lb = b.name_e - b.name_b;
            int c = std::memcmp(sa.data() + a.name_b, sb.data() + b.name_b, std::min(la, lb));
            if (c == 0 && la != lb) c = la < lb ? -1 : 1;
            if (c != 0) return c < 0;
            const size_t ra = a.end - a.begin, rb = b.end - b.begin;
            c = std::memcmp(sa.data() + a.begin, sb.data() + b.begin, std::min(ra, rb));
            return c != 0 ? c < 0 : ra < rb;
        };
        std::sort(order.begin(), order.end(), key_less);
        std::string text;
        for (size_t k = 0; k < n; ++k) {
            const Placed &a = placed[order[k]];
            text.append(kept[a.block], a.begin, a.end - a.begin);
            if (text.size() > ((size_t)32 << 20)) {
                out.write(text.data(), (std::streamsize)text.size());
                text.clear();
            }
        }
        out.write(text.data(), (std::streamsize)text.size());
    }
    if (!out || (feature_path && !fout)) throw std::runtime_error("ERROR: Could not write the output file.");
}

}  // namespace vsc_merge
