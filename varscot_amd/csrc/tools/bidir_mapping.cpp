// bidir_mapping - MI355X drop-in for VARSCOT_pipeline/read_mapping/bidir_mapping.cpp.
// Same flags (:196-216), same stdout lines (:265,269), same errors / exit codes (:218-220,234-238,
// 301-305), same header-less SAM output in the same record order (:88-123,167-187,307-308).
// Host C++ only: the search runs in libvarscot_hip.so through the C ABI; there is no CPU search here.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <fstream>

#include "vsc_host.hpp"

using namespace vsc_host;

int main(int argc, char **argv)
{
    std::vector<Option> opts = {
        {'G', "genome", "Path to genome fasta file", true},
        {'I', "index", "Path to the indexed genome", true},
        {'R', "reads", "Path to the reads (have to be Dna4)", true},
        {'M', "mismatches", "Number of allowed mismatches", true},
        {'T', "threads", "Number of threads (accepted for compatibility; the GPU replaces the OpenMP loop)", false},
        {'O', "output", "Path to output SAM file", false},
        {'P', "pam", "Additional non-canonical PAM that should be allowed for off-target search besides (N)GG and (N)GA (default).", false},
        {'D', "device", "HIP device index (default 0), or a comma-separated list (0,1,2,...): the genome is sharded over these "
                        "devices, each searches all reads, one RCCL gather brings the hits to the first", false},
        {'S', "md-style", "0 = SAM-spec MD strings (default), 1 = no zeros between adjacent mismatches", false},
    };
    const int pr = parse_args(argc, argv, opts, "Read mapping",
                              "Read mapper for CRISPR-Cas9 off-targets. Only supports Dna4 reads (everything else than "
                              "ACGT will be converted to A). All reads must have length 23.");
    if (pr) return pr == 1;
    const std::string genome_path = opts[0].value, index_prefix = opts[1].value, reads_path = opts[2].value;
    const std::string out_path = opts[5].value, pam = opts[6].value;
    if (!has_extension(genome_path, {"fa", "fasta"}) || !has_extension(reads_path, {"fa", "fasta"}) ||
        (opts[5].set && !has_extension(out_path, {"sam", "bam"}))) {
        std::fprintf(stderr, "%s: genome and reads must be .fa/.fasta files, the output a .sam/.bam file\n", argv[0]);
        return 1;
    }
    char *end = nullptr;
    const long mm = std::strtol(opts[3].value.c_str(), &end, 10);
    if (end == opts[3].value.c_str() || *end) {
        std::fprintf(stderr, "%s: the given value '%s' cannot be casted to integer\n", argv[0], opts[3].value.c_str());
        return 1;
    }
    if (mm < 0 || mm > 8) {  // bidir_mapping.cpp:234-238
        std::fprintf(stderr, "Error: Maximum number of mismatches must lie between 0 and 8.\n");
        return 1;
    }
    std::vector<int> devices;  // -D 0 | -D 0,1,2,3 (an id may repeat: several shards on one device)
    {
        const std::string d = opts[7].set ? opts[7].value : "0";
        size_t b = 0;
        for (;;) {
            const size_t e = d.find(',', b);
            const std::string item = d.substr(b, e == std::string::npos ? std::string::npos : e - b);
            char *iend = nullptr;
            const long v = std::strtol(item.c_str(), &iend, 10);
            if (item.empty() || *iend || v < 0) {
                std::fprintf(stderr, "%s: bad device list '%s'\n", argv[0], d.c_str());
                return 1;
            }
            devices.push_back((int)v);
            if (e == std::string::npos) break;
            b = e + 1;
        }
    }
    const int md_style = opts[8].set ? std::atoi(opts[8].value.c_str()) : 0;

    vsc_ctx *ctx = nullptr;
    vsc_genome *genome = nullptr;
    vsc_multi *multi = nullptr;
    vsc_multi_genome *mgenome = nullptr;
    vsc_hits *hits = nullptr;
    int rc = 1;
    try {
        const auto reads = read_fasta(reads_path);
        std::printf("Reads loaded (total: %zu).\n", reads.size());
        std::vector<uint64_t> codes(reads.size());
        for (size_t i = 0; i < reads.size(); ++i) {
            if (reads[i].seq.size() != VSC_READ_LEN)
                throw std::runtime_error("read '" + reads[i].id + "' is not 23 nt long (VARSCOT searches 20 nt + PAM)");
            codes[i] = vsc_pack_guide(reads[i].seq.c_str());
        }
        const PackedIndex ix = read_index(index_prefix);
        int st;
        if (devices.size() == 1) {
            st = vsc_ctx_create(devices[0], &ctx);
            if (st != VSC_OK) throw std::runtime_error(st == VSC_ERR_NODEVICE ? "no HIP device available (there is no CPU fallback)" : "could not create the device context");
            st = vsc_genome_load(ctx, ix.hi.data(), ix.lo.data(), ix.nm.data(), 0, ix.hi.size(), ix.hi.size(), ix.contigs.data(),
                                 (uint32_t)ix.contigs.size(), &genome);
            if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
            // a seed index written by `bidir_index -S` is taken if it fits this genome and the default PAM set;
            // otherwise (absent, stale, other PAM) the search builds the index itself
            if (pam.size() != 2 && std::ifstream(seed_index_path(index_prefix)).good() &&
                vsc_genome_index_load(ctx, genome, seed_index_path(index_prefix).c_str()) != VSC_OK)
                std::fprintf(stderr, "%s: %s - building the seed index instead\n", argv[0], vsc_last_error(ctx));
        } else {
            st = vsc_multi_create(devices.data(), (int)devices.size(), &multi);
            if (st != VSC_OK) throw std::runtime_error(st == VSC_ERR_NODEVICE ? "no HIP device available (there is no CPU fallback)" : "could not create the device contexts");
            st = vsc_multi_genome_load(multi, ix.hi.data(), ix.lo.data(), ix.nm.data(), ix.hi.size(), ix.contigs.data(),
                                       (uint32_t)ix.contigs.size(), &mgenome);
            if (st != VSC_OK) throw std::runtime_error(vsc_multi_last_error(multi));
            ctx = vsc_multi_ctx(multi, 0);  // owns the merged result
        }
        std::printf("Index loaded.\n");

        vsc_search_params p{};
        p.max_mismatches = (uint32_t)mm;
        if (pam.size() == 2) {  // a PAM of any other length can never equal a 2-base window slice (:71-76)
            p.has_extra_pam = 1;
            p.extra_pam[0] = pam[0];
            p.extra_pam[1] = pam[1];
        }
        if (multi) {
            st = vsc_multi_search(multi, mgenome, codes.data(), (uint32_t)codes.size(), &p, &hits);
            if (st != VSC_OK) throw std::runtime_error(vsc_multi_last_error(multi));
        } else {
            st = vsc_search(ctx, genome, codes.data(), (uint32_t)codes.size(), &p, &hits);
            if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
        }
        const uint64_t n = vsc_hits_count(hits);
        const vsc_hit *h = nullptr;
        st = vsc_hits_data(hits, &h);
        if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
        std::vector<uint64_t> order(n);
        std::vector<uint8_t> secondary(n);
        vsc_sam_order(h, n, order.data(), secondary.data());

        std::ofstream out(out_path);
        if (!out.is_open()) {  // :301-305
            std::fprintf(stderr, "ERROR: Could not open output path.\n");
        } else {
            std::string text;
            std::vector<std::string> seq(reads.size());
            for (size_t i = 0; i < reads.size(); ++i) seq[i] = dna4(reads[i].seq);
            char window[VSC_READ_LEN + 1] = {0};
            for (uint64_t i = 0; i < n; ++i) {
                const vsc_hit &r = h[order[i]];
                vsc_unpack_bases(ix.hi.data(), ix.lo.data(), ix.nm.data(), ix.contigs[r.contig].offset + r.pos, VSC_READ_LEN,
                                 window);
                append_sam_line(text, reads[r.guide].id, ix.names[r.contig], r, secondary[i] != 0, seq[r.guide], window,
                                md_style);
                if (text.size() > (1u << 22)) {
                    out << text;
                    text.clear();
                }
            }
            out << text;
            out.close();
            rc = 0;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        rc = 1;
    }
    vsc_hits_free(hits);
    if (multi) {
        vsc_multi_genome_free(mgenome);
        vsc_multi_destroy(multi);
    } else {
        vsc_genome_free(genome);
        vsc_ctx_destroy(ctx);
    }
    return rc;
}
