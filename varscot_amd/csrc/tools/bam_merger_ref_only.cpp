// bam_merger_ref_only - drop-in for VARSCOT_pipeline/variant_processing/bam_merger_ref_only.cpp
// (processRefOnly, merge_output_bam.h:485-720): reference SAM + on-target BED + genome FASTA + on-target
// activity -> TSV (MIT score, last argument 0) or TSV + feature matrix (1).  Scores are computed by
// libvarscot_hip.so on the GPU (vsc_score_pairs).
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
    if (argc != 10) {
        std::cerr << "USAGE: bam_merger RESULT_MERGED.txt FEATURE_MATRIX.txt RESULT_REF.bam ONTARGETS.bed GENOME.fa "
                     "TUSCAN_REGRESSION.txt NUMMISMATCHES SEQLENGTH MIT\n";
        return 1;
    }
    unsigned vals[3];
    for (int i = 0; i < 3; ++i)
        if (!to_unsigned(argv[7 + i], &vals[i])) {
            std::cerr << "ERROR: Cannot cast " << argv[7 + i] << " into an unsigned.\n";
            return 1;
        }
    const unsigned mit = vals[2];
    try {
        const Genome genome(argv[5]);
        std::cout << "Read reference BAM file" << std::endl;
        const auto hits = read_sam(argv[3], genome);
        std::map<std::string, OffTarget> on;
        std::map<std::string, unsigned> count;
        read_ontargets(argv[4], genome, on, count);
        const auto activity = read_tuscan(argv[6]);
        std::vector<const OffTarget *> rows;
        for (const auto &h : hits)
            if (!same(h, on.at(h.target))) rows.push_back(&h);  // merge_output_bam.h:534,661
        const std::string feature_path = argv[2];
        write_outputs(argv[1], mit == 0 ? nullptr : &feature_path, false, rows, on, count, activity, 0);
        std::cout << "Writing reference output finished." << std::endl;
    } catch (const std::exception &e) {
        std::cout << e.what() << std::endl;
        return 1;
    }
    return 0;
}
