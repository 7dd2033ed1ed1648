// vsc_host.hpp - host-side helpers shared by the command-line tools: FASTA input, the on-disk packed
// genome ("index"), SAM text.  Plain C++17, no device code; everything GPU goes through the C ABI
// (include/varscot_hip.h).
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
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
// magic "VSCIDX02" | u64 n_contigs | u64 n_words | u64 size and i64 mtime (ns) of the FASTA it was packed from (0: unknown) |
// contig table | names (u32 length + bytes) | zero padding to a multiple of 8 bytes | hi plane | lo plane | N plane.
// ("VSCIDX01", rounds 1-3: no source fields, no padding - still read.)  Replaces the SeqAn index files of
// read_mapping/bidir_index.cpp:45-47 AND the .fai index the reference's other stages open next to the FASTA
// (variant_processing/extract_fasta_ontargets.h:33-76, filter_output_bam.h:399, write_fasta.h:245-271): every tool takes
// its reference bases from this file when it is there and belongs to the FASTA it was given (PackedView below).
struct PackedIndex {
    std::vector<vsc_contig> contigs;
    std::vector<std::string> names;
    std::vector<uint32_t> hi, lo, nm;
    uint64_t src_size = 0;   // the FASTA this was packed from: its size ...
    int64_t src_mtime = 0;   // ... and modification time in nanoseconds (0: unknown)
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

inline bool file_stamp(const std::string &path, uint64_t *size, int64_t *mtime_ns)
{
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return false;
    *size = (uint64_t)st.st_size;
    *mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}

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
    // packed on all host threads: vsc_pack_bases only writes the bits of the range it is given, so whole words can be
    // handed out freely
    const unsigned n_threads = std::min<unsigned>(host_threads(), (unsigned)std::max<size_t>(1, recs.size()));
    if (n_threads <= 1 || recs.size() < 2) {
        for (size_t i = 0; i < recs.size(); ++i)
            vsc_pack_bases(recs[i].seq.data(), recs[i].seq.size(), ix.contigs[i].offset, ix.hi.data(), ix.lo.data(), ix.nm.data());
    } else {
        // whole words belong to one worker: a contig is cut at word boundaries, the (at most two) partial words at its ends
        // are packed afterwards, one contig after the other
        struct Piece { size_t rec; uint64_t b, e; };
        std::vector<Piece> whole, edge;
        for (size_t i = 0; i < recs.size(); ++i) {
            const uint64_t o = ix.contigs[i].offset, n = recs[i].seq.size();
            const uint64_t wb = (o + 31) / 32 * 32, we = (o + n) / 32 * 32;
            if (we <= wb) {
                if (n) edge.push_back({i, 0, n});
                continue;
            }
            if (wb > o) edge.push_back({i, 0, wb - o});
            if (o + n > we) edge.push_back({i, we - o, n});
            const uint64_t step = (uint64_t)1 << 24;  // 16 M bases per piece
            for (uint64_t b = wb; b < we; b += step) whole.push_back({i, b - o, std::min(we, b + step) - o});
        }
        std::vector<std::thread> pool;
        std::atomic<size_t> next{0};
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back([&] {
                for (size_t k; (k = next.fetch_add(1)) < whole.size();) {
                    const Piece &pc = whole[k];
                    vsc_pack_bases(recs[pc.rec].seq.data() + pc.b, pc.e - pc.b, ix.contigs[pc.rec].offset + pc.b, ix.hi.data(), ix.lo.data(),
                                   ix.nm.data());
                }
            });
        for (auto &th : pool) th.join();
        for (const Piece &pc : edge)
            vsc_pack_bases(recs[pc.rec].seq.data() + pc.b, pc.e - pc.b, ix.contigs[pc.rec].offset + pc.b, ix.hi.data(), ix.lo.data(), ix.nm.data());
    }
    return ix;
}

inline void write_index(const std::string &prefix, const PackedIndex &ix)
{
    std::ofstream out(index_path(prefix), std::ios::binary);
    if (!out) throw std::runtime_error("Could not open " + index_path(prefix) + " for writing");
    const uint64_t nc = ix.contigs.size(), nw = ix.hi.size();
    out.write("VSCIDX02", 8);
    out.write((const char *)&nc, 8);
    out.write((const char *)&nw, 8);
    out.write((const char *)&ix.src_size, 8);
    out.write((const char *)&ix.src_mtime, 8);
    out.write((const char *)ix.contigs.data(), (std::streamsize)(nc * sizeof(vsc_contig)));
    uint64_t name_bytes = 0;
    for (const auto &n : ix.names) {
        const uint32_t l = (uint32_t)n.size();
        out.write((const char *)&l, 4);
        out.write(n.data(), l);
        name_bytes += 4 + l;
    }
    const char zeros[8] = {0};
    out.write(zeros, (std::streamsize)((8 - name_bytes % 8) % 8));  // the planes start 8-byte aligned (mapped readers)
    out.write((const char *)ix.hi.data(), (std::streamsize)(nw * 4));
    out.write((const char *)ix.lo.data(), (std::streamsize)(nw * 4));
    out.write((const char *)ix.nm.data(), (std::streamsize)(nw * 4));
    if (!out) throw std::runtime_error("Write error on " + index_path(prefix));
}

// A packed genome file mapped read-only: nothing is read until it is touched - the few regions fasta_writer or a merger
// asks for cost a few pages, not a pass over 3 GB of FASTA text.  (Files of the older layout are read into memory.)
class PackedView {
public:
    std::vector<vsc_contig> contigs;
    std::vector<std::string> names;
    const uint32_t *hi = nullptr, *lo = nullptr, *nm = nullptr;
    uint64_t n_words = 0;
    uint64_t src_size = 0;
    int64_t src_mtime = 0;

    PackedView() = default;
    PackedView(const PackedView &) = delete;
    PackedView &operator=(const PackedView &) = delete;
    ~PackedView()
    {
        if (map_) ::munmap(map_, map_len_);
    }
    // false: no such file; throws on a file that is there but is not a packed genome
    bool open(const std::string &path)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0 || st.st_size < 24) {
            ::close(fd);
            throw std::runtime_error("Not a packed genome: " + path);
        }
        map_len_ = (size_t)st.st_size;
        map_ = ::mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (map_ == MAP_FAILED) {
            map_ = nullptr;
            throw std::runtime_error("Could not map " + path);
        }
        const char *p = (const char *)map_, *const end = p + map_len_;
        const bool v2 = std::memcmp(p, "VSCIDX02", 8) == 0;
        if (!v2 && std::memcmp(p, "VSCIDX01", 8) != 0) throw std::runtime_error("Not a packed genome: " + path);
        auto need = [&](uint64_t bytes) {
            if ((uint64_t)(end - p) < bytes) throw std::runtime_error("Truncated index " + path);
        };
        uint64_t nc = 0;
        need(v2 ? 40 : 24);
        std::memcpy(&nc, p + 8, 8);
        std::memcpy(&n_words, p + 16, 8);
        if (v2) {
            std::memcpy(&src_size, p + 24, 8);
            std::memcpy(&src_mtime, p + 32, 8);
        }
        p += v2 ? 40 : 24;
        if (nc > (1ull << 32)) throw std::runtime_error("Truncated index " + path);
        need(nc * sizeof(vsc_contig));
        contigs.resize(nc);
        std::memcpy(contigs.data(), p, nc * sizeof(vsc_contig));
        p += nc * sizeof(vsc_contig);
        names.resize(nc);
        uint64_t name_bytes = 0;
        for (auto &n : names) {
            uint32_t l = 0;
            need(4);
            std::memcpy(&l, p, 4);
            need(4 + (uint64_t)l);
            n.assign(p + 4, l);
            p += 4 + l;
            name_bytes += 4 + l;
        }
        if (v2) p += (8 - name_bytes % 8) % 8;
        if (n_words > (uint64_t)(end - p) / 12) throw std::runtime_error("Truncated index " + path);
        // (regions are clamped to their contig's length, bases() trusts the table: a contig must lie inside the planes)
        for (const vsc_contig &c : contigs)
            if ((uint64_t)c.offset > n_words * 32 || (uint64_t)c.length > n_words * 32 - (uint64_t)c.offset)
                throw std::runtime_error("Corrupt contig table in " + path);
        if (((uintptr_t)p & 3u) == 0) {
            hi = (const uint32_t *)p;
            lo = hi + n_words;
            nm = lo + n_words;
        } else {  // an older file whose names left the planes unaligned: one copy
            own_.resize(n_words * 3);
            std::memcpy(own_.data(), p, n_words * 12);
            hi = own_.data();
            lo = hi + n_words;
            nm = lo + n_words;
        }
        return true;
    }
    // does this file belong to the FASTA at `fasta` (same size and modification time as when it was packed)?
    bool matches(const std::string &fasta) const
    {
        uint64_t size = 0;
        int64_t mtime = 0;
        return src_size != 0 && file_stamp(fasta, &size, &mtime) && size == src_size && mtime == src_mtime;
    }
    std::string bases(uint32_t contig, uint64_t begin, uint64_t n) const  // upper-case ACGT, everything else N
    {
        std::string out(n, 'N');
        if (n) vsc_unpack_bases(hi, lo, nm, contigs[contig].offset + begin, n, &out[0]);
        return out;
    }

private:
    void *map_ = nullptr;
    size_t map_len_ = 0;
    std::vector<uint32_t> own_;
};

inline PackedIndex read_index(const std::string &prefix)
{
    PackedView v;
    if (!v.open(index_path(prefix))) throw std::runtime_error("Could not open index " + index_path(prefix));
    PackedIndex ix;
    ix.contigs = v.contigs;
    ix.names = v.names;
    ix.hi.assign(v.hi, v.hi + v.n_words);
    ix.lo.assign(v.lo, v.lo + v.n_words);
    ix.nm.assign(v.nm, v.nm + v.n_words);
    ix.src_size = v.src_size;
    ix.src_mtime = v.src_mtime;
    return ix;
}

// The packed genome that stands for the FASTA at `fasta`, if there is one: the prefix in the environment variable `env`
// (the driver exports the -i prefix as VARSCOT_PACKED_GENOME, the SNP genome's as VARSCOT_PACKED_SNP_GENOME), else
// <fasta>.vsc, else the FASTA's name with .vsc in place of its extension - taken only if the file says it was packed from
// a FASTA of this size and modification time.  Returns false when there is none (the caller parses the FASTA).
inline bool open_packed_for(const std::string &fasta, const char *env, PackedView &view)
{
    std::vector<std::string> tries;
    const char *e = env ? std::getenv(env) : nullptr;
    if (e && *e) tries.push_back(index_path(e));
    tries.push_back(fasta + ".vsc");
    const size_t dot = fasta.rfind('.');
    if (dot != std::string::npos && fasta.find('/', dot) == std::string::npos) tries.push_back(fasta.substr(0, dot) + ".vsc");
    const char *trace = std::getenv("VARSCOT_TRACE");  // tests: say on stderr where the reference bases come from
    for (const auto &path : tries) {
        PackedView v;
        try {
            if (!v.open(path)) continue;
            if (!v.matches(fasta)) {
                if (trace) std::fprintf(stderr, "[varscot] %s was not packed from %s as it is now: not used\n", path.c_str(), fasta.c_str());
                continue;
            }
        } catch (const std::exception &) {
            continue;
        }
        if (view.open(path)) {  // (`view` is a fresh object: nothing mapped yet)
            if (trace) std::fprintf(stderr, "[varscot] reference bases of %s from the packed genome %s\n", fasta.c_str(), path.c_str());
            return true;
        }
    }
    if (trace) std::fprintf(stderr, "[varscot] reference bases of %s from the FASTA text\n", fasta.c_str());
    return false;
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
