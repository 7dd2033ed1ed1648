// bidir_index - packs a genome FASTA into the resident-plane "index" of libvarscot_hip.
// Same command line as VARSCOT_pipeline/read_mapping/bidir_index.cpp:19-24 (-G genome, -I prefix) and
// the same two stdout lines (:42,49); the SeqAn FM-index files are replaced by <prefix>.vsc.
#include <cstdio>
#include <exception>

#include "vsc_host.hpp"

using namespace vsc_host;

int main(int argc, char **argv)
{
    std::vector<Option> opts = {
        {'G', "genome", "Path to the genome (.fa, .fasta, .fastq)", true},
        {'I', "index", "Path to the index", true},
    };
    const int pr = parse_args(argc, argv, opts, "VARSCOT - Index Creation",
                              "Packs a multi-sequence FASTA file (A, C, G, T, N) into the bit planes the MI355X "
                              "search keeps resident. The FASTA file may not contain more than 4 giga bases in total.");
    if (pr) return pr == 1;
    const std::string genome = opts[0].value, prefix = opts[1].value;
    if (!has_extension(genome, {"fa", "fasta", "fastq"})) {
        std::fprintf(stderr, "%s: the genome must be a .fa, .fasta or .fastq file\n", argv[0]);
        return 1;
    }
    try {
        const auto recs = read_fasta(genome);
        std::printf("Number of sequences: %zu\n", recs.size());
        uint64_t total = 0;
        for (const auto &r : recs) total += r.seq.size() + 1;
        if (total >= (1ull << 32) - 8192) {
            std::fprintf(stderr, "ERROR: the FASTA file may not contain more than 4 giga bases in total.\n");
            return 1;
        }
        write_index(prefix, pack_records(recs));
        std::printf("Index created successfully\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
