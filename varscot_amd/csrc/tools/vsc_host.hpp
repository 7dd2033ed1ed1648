// vsc_host.hpp - host-side helpers shared by the command-line tools: FASTA input, the on-disk packed
// genome ("index"), SAM text.  Plain C++17, no device code; everything GPU goes through the C ABI
// (include/varscot_hip.h).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "varscot_hip.h"

namespace vsc_host {

struct FastaRecord {
    std::string id;   // full header line without '>' (what SeqAn's readRecord returns)
    std::string seq;
};

// Multi-FASTA reader (line-wrapped sequences, CR/LF tolerant, blank lines ignored).
inline std::vector<FastaRecord> read_fasta(const std::string &path)
{
    // whole lines are appended at once (memchr / append): a 3 GB genome parses at disk speed, not at one
    // push_back per base
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Could not open " + path);
    std::vector<FastaRecord> out;
    std::vector<char> buf(1 << 24);
    std::string carry;  // an incomplete line at the end of the previous block
    auto take_line = [&](const char *p, size_t n) {
        while (n && (p[n - 1] == '\r' || p[n - 1] == '\n')) --n;
        if (n == 0) return;
        if (p[0] == '>') {
            out.push_back({std::string(p + 1, n - 1), std::string()});
        } else if (!out.empty()) {
            std::string &seq = out.back().seq;
            if (!std::memchr(p, ' ', n) && !std::memchr(p, '\t', n)) {
                seq.append(p, n);
            } else {
                for (size_t i = 0; i < n; ++i)
                    if (p[i] != ' ' && p[i] != '\t') seq.push_back(p[i]);
            }
        }
    };
    for (;;) {
        const size_t got = std::fread(buf.data(), 1, buf.size(), f);
        if (got == 0) break;
        const char *p = buf.data(), *end = p + got;
        while (p < end) {
            const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
            if (!nl) {
                carry.append(p, (size_t)(end - p));
                break;
            }
            if (!carry.empty()) {
                carry.append(p, (size_t)(nl - p));
                take_line(carry.data(), carry.size());
                carry.clear();
            } else {
                take_line(p, (size_t)(nl - p));
            }
            p = nl + 1;
        }
    }
    if (!carry.empty()) take_line(carry.data(), carry.size());
    std::fclose(f);
    return out;
}

// ---- packed genome on disk: <prefix>.vsc -------------------------------------------------------
// magic "VSCIDX01" | u64 n_contigs | u64 n_words | contig table | names (u32 length + bytes) |
// hi plane | lo plane | N plane.  Replaces the SeqAn index files of read_mapping/bidir_index.cpp:45-47.
struct PackedIndex {
    std::vector<vsc_contig> contigs;
    std::vector<std::string> names;
    std::vector<uint32_t> hi, lo, nm;
};

// worker threads for host-side text work: OMP_NUM_THREADS if the driver set it (VARSCOT:257), else all cores
inline unsigned host_threads()
{
    const char *e = std::getenv("OMP_NUM_THREADS");
    const long v = e ? std::strtol(e, nullptr, 10) : 0;
    const unsigned hw = std::thread::hardware_concurrency();
    return (unsigned)std::max<long>(1, std::min<long>(v > 0 ? v : (long)(hw ? hw : 1), 64));
}

inline std::string index_path(const std::string &prefix) { return prefix + ".vsc"; }
inline std::string seed_index_path(const std::string &prefix) { return prefix + ".vsi"; }  // optional: bidir_index -S

inline PackedIndex pack_records(const std::vector<FastaRecord> &recs)
{
    PackedIndex ix;
    std::vector<uint32_t> len(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        if (recs[i].seq.size() >= (1ull << 32)) throw std::runtime_error("contig longer than 4 Gbases");
        len[i] = (uint32_t)recs[i].seq.size();
        ix.names.push_back(recs[i].id);
    }
    ix.contigs.resize(recs.size());
    uint64_t n_words = vsc_layout_contigs(len.data(), (uint32_t)len.size(), ix.contigs.data());
    if (n_words == 0) n_words = 1;
    ix.hi.resize(n_words);
    ix.lo.resize(n_words);
    ix.nm.resize(n_words);
    vsc_planes_init(ix.hi.data(), ix.lo.data(), ix.nm.data(), n_words);
    for (size_t i = 0; i < recs.size(); ++i)
        vsc_pack_bases(recs[i].seq.data(), recs[i].seq.size(), ix.contigs[i].offset, ix.hi.data(), ix.lo.data(),
                       ix.nm.data());
    return ix;
}

inline void write_index(const std::string &prefix, const PackedIndex &ix)
{
    std::ofstream out(index_path(prefix), std::ios::binary);
    if (!out) throw std::runtime_error("Could not open " + index_path(prefix) + " for writing");
    const uint64_t nc = ix.contigs.size(), nw = ix.hi.size();
    out.write("VSCIDX01", 8);
    out.write((const char *)&nc, 8);
    out.write((const char *)&nw, 8);
    out.write((const char *)ix.contigs.data(), (std::streamsize)(nc * sizeof(vsc_contig)));
    for (const auto &n : ix.names) {
        const uint32_t l = (uint32_t)n.size();
        out.write((const char *)&l, 4);
        out.write(n.data(), l);
    }
    out.write((const char *)ix.hi.data(), (std::streamsize)(nw * 4));
    out.write((const char *)ix.lo.data(), (std::streamsize)(nw * 4));
    out.write((const char *)ix.nm.data(), (std::streamsize)(nw * 4));
    if (!out) throw std::runtime_error("Write error on " + index_path(prefix));
}

inline PackedIndex read_index(const std::string &prefix)
{
    std::ifstream in(index_path(prefix), std::ios::binary);
    if (!in) throw std::runtime_error("Could not open index " + index_path(prefix));
    char magic[8];
    uint64_t nc = 0, nw = 0;
    in.read(magic, 8);
    in.read((char *)&nc, 8);
    in.read((char *)&nw, 8);
    if (!in || std::memcmp(magic, "VSCIDX01", 8) != 0) throw std::runtime_error("Not a packed genome: " + index_path(prefix));
    PackedIndex ix;
    ix.contigs.resize(nc);
    in.read((char *)ix.contigs.data(), (std::streamsize)(nc * sizeof(vsc_contig)));
    ix.names.resize(nc);
    for (auto &n : ix.names) {
        uint32_t l = 0;
        in.read((char *)&l, 4);
        n.resize(l);
        in.read(&n[0], l);
    }
    ix.hi.resize(nw);
    ix.lo.resize(nw);
    ix.nm.resize(nw);
    in.read((char *)ix.hi.data(), (std::streamsize)(nw * 4));
    in.read((char *)ix.lo.data(), (std::streamsize)(nw * 4));
    in.read((char *)ix.nm.data(), (std::streamsize)(nw * 4));
    if (!in) throw std::runtime_error("Truncated index " + index_path(prefix));
    return ix;
}

// ---- SAM text (read_mapping/bidir_mapping.cpp:88-123) -----------------------------
// MD:Z value from the mismatch mask and the window's bases (reference bases at the mismatches).
// md_style 0: SAM-spec zeros; 1: no zeros between adjacent mismatches / at the ends.
inline std::string md_string(uint32_t mask, const char *window, int md_style)
{
    std::string md;
    int run = 0;
    for (int i = 0; i < VSC_READ_LEN; ++i) {
        if (!((mask >> i) & 1u)) {
            ++run;
        } else {
            if (run > 0 || md_style == 0) md += std::to_string(run);
            md.push_back(window[i]);
            run = 0;
        }
    }
    if (run > 0 || md_style == 0) md += std::to_string(run);
    return md;
}

// SEQ column: the read as given, with SeqAn's Dna conversion (non-ACGT -> A, upper case)
inline std::string dna4(const std::string &s)
{
    std::string o(s);
    for (auto &c : o) {
        switch (c) {
        case 'A': case 'a': c = 'A'; break;
        case 'C': case 'c': c = 'C'; break;
        case 'G': case 'g': c = 'G'; break;
        case 'T': case 't': c = 'T'; break;
        default: c = 'A';
        }
    }
    return o;
}

inline void append_sam_line(std::string &out, const std::string &qname, const std::string &rname, const vsc_hit &h,
                            bool secondary, const std::string &seq, const char *window, int md_style)
{
    const unsigned flag = (VSC_HIT_STRAND(h.info) ? 16u : 0u) | (secondary ? 256u : 0u);
    out += qname;
    out += '\t';
    out += std::to_string(flag);
    out += '\t';
    out += rname;
    out += '\t';
    out += std::to_string(h.pos + 1);
    out += "\t255\t23M\t*\t0\t0\t";
    out += seq;
    out += "\tIIIIIIIIIIIIIIIIIIIIIII\tNM:i:";
    out += std::to_string(VSC_HIT_NM(h.info));
    out += "\tMD:Z:";
    out += md_string(VSC_HIT_MASK(h.info), window, md_style);
    out += '\n';
}

// ---- tiny argv parser with the behaviour the tools need from SeqAn's ArgumentParser ---------------
struct Option {
    char short_name;
    const char *long_name;
    const char *help;
    bool required;
    bool has_value = true;
    std::string value;
    bool set = false;
};

// returns 0 = ok, 1 = parse error (message printed), 2 = help printed
inline int parse_args(int argc, char **argv, std::vector<Option> &opts, const char *title, const char *description)
{
    auto usage = [&]() {
        std::printf("%s\n\n%s\n\nOPTIONS\n", title, description);
        for (auto &o : opts) std::printf("  -%c, --%s %s\n        %s\n", o.short_name, o.long_name, o.has_value ? "ARG" : "", o.help);
        std::printf("  -h, --help\n        Display this help message.\n");
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            usage();
            return 2;
        }
        Option *hit = nullptr;
        std::string inline_value;
        bool has_inline = false;
        for (auto &o : opts) {
            if ((a.size() == 2 && a[0] == '-' && a[1] == o.short_name) || a == std::string("--") + o.long_name) hit = &o;
            const std::string pre = std::string("--") + o.long_name + "=";
            if (a.compare(0, pre.size(), pre) == 0) {
                hit = &o;
                inline_value = a.substr(pre.size());
                has_inline = true;
            }
        }
        if (!hit) {
            std::fprintf(stderr, "%s: illegal option -- %s\n", argv[0], a.c_str());
            return 1;
        }
        if (hit->has_value) {
            if (has_inline) {
                hit->value = inline_value;
            } else {
                if (i + 1 >= argc) {
                    std::fprintf(stderr, "%s: option requires an argument -- %s\n", argv[0], a.c_str());
                    return 1;
                }
                hit->value = argv[++i];
            }
        }
        hit->set = true;
    }
    for (auto &o : opts)
        if (o.required && !o.set) {
            std::fprintf(stderr, "%s: Missing value for option: -%c\n", argv[0], o.short_name);
            return 1;
        }
    return 0;
}

inline bool has_extension(const std::string &path, std::initializer_list<const char *> exts)
{
    const size_t dot = path.rfind('.');
    if (dot == std::string::npos) return false;
    const std::string e = path.substr(dot + 1);
    for (const char *x : exts)
        if (e == x) return true;
    return false;
}

}  // namespace vsc_host
