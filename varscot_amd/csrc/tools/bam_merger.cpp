// bam_merger - drop-in for VARSCOT_pipeline/variant_processing/bam_merger.cpp (mergeResults,
// merge_output_bam.h:46-460): merges the mappings against the reference and against the SNP genome,
// drops on-targets, reference hits shadowed by a variant window and duplicates, restores variant
// coordinates, and writes the TSV (+ feature matrix).  Scores come from the GPU (vsc_score_pairs).
#include "merge_host.hpp"

using namespace vsc_merge;

static bool to_unsigned(const char *s, unsigned *out)
{
    char *end = nullptr;
    if (!*s || *s == '-') return false;
    unsigned long v = std::strtoul(s, &end, 10);
    if (*end) return false;
    *out = (unsigned)v;
    return true;
}

int main(int argc, char **argv)
{
    if (argc != 13) {
        std::cerr << "USAGE: bam_merger RESULT_MERGED.txt FEATURE_MATRIX.txt RESULT_REF.bam RESULT_SNP.bam ONTARGETS.bed "
                     "GENOME.fa VARIANT_GENOME.fa TUSCAN_REGRESSION.txt NUMMISMATCHES SEQLENGTH THREADS MIT\n";
        return 1;
    }
    unsigned vals[4];
    for (int i = 0; i < 4; ++i)
        if (!to_unsigned(argv[9 + i], &vals[i])) {
            std::cerr << "ERROR: Cannot cast " << argv[9 + i] << " into an unsigned.\n";
            return 1;
        }
    const unsigned seq_len = vals[1], mit = vals[3];
    try {
        const Genome snp(argv[7], "VARSCOT_PACKED_SNP_GENOME");
        const Genome ref(argv[6]);
        const WindowIndex windows(snp);  // getSnpInfoTable + sortSnpRegionsByChr
        std::map<std::string, OffTarget> on;
        std::map<std::string, unsigned> count;
        read_ontargets(argv[5], ref, on, count);

        std::cout << "Process reference off-targets" << std::endl;
        const auto ref_hits = read_sam(argv[3], ref);
        std::vector<const OffTarget *> rows;
        for (const auto &h : ref_hits)  // filterRefAlignment, filter_output_bam.h:70-124
            if (!same(h, on.at(h.target)) && !windows.shadows(h.chr, h.pos, seq_len)) rows.push_back(&h);

        std::cout << "Process variant off-targets" << std::endl;
        auto snp_hits = read_sam(argv[4], snp);
        for (size_t i = 0; i < snp_hits.size(); ++i) {  // filterSnpAlignment, :279-317
            OffTarget &h = snp_hits[i];
            const auto id = split_id(h.chr);
            h.chr = id[0];
            h.pos = h.pos + (uint32_t)c_atoi(id.size() > 1 ? id[1] : "0");
            snp_type(h.snp_type, id, h.pos, seq_len);
            bool valid = !same(h, on.at(h.target));
            if (i > 0 && same(h, snp_hits[i - 1])) valid = false;
            if (valid) rows.push_back(&h);
        }
        const auto activity = read_tuscan(argv[8]);
        const std::string feature_path = argv[2];
        write_outputs(argv[1], mit == 0 ? nullptr : &feature_path, true, rows, on, count, activity, 0);
        std::cout << "Merging output files finished" << std::endl;
    } catch (const std::exception &e) {
        std::cout << e.what() << std::endl;
        return 1;
    }
    return 0;
}
