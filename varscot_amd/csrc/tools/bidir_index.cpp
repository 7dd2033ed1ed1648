// bidir_index - packs a genome FASTA into the resident-plane "index" of libvarscot_hip.
// Same command line as VARSCOT_pipeline/read_mapping/bidir_index.cpp:19-24 (-G genome, -I prefix) and
// the same two stdout lines (:42,49); the SeqAn FM-index files are replaced by <prefix>.vsc.
// -S (additional) also builds the seed index on device 0 and writes it to <prefix>.vsi, which bidir_mapping loads
// instead of building it (36 bytes per PAM-valid window; worth it on small genomes and slow devices only - at 3 Gbp
// rebuilding on the device takes 0.3 s, reading 27 GB back does not).
#include <cstdio>
#include <exception>
#include <stdexcept>

#include "vsc_host.hpp"

using namespace vsc_host;

int main(int argc, char **argv)
{
    std::vector<Option> opts = {
        {'G', "genome", "Path to the genome (.fa, .fasta, .fastq)", true},
        {'I', "index", "Path to the index", true},
        {'S', "seed-index", "Also build the seed index (needs the GPU) and write it to <index>.vsi", false, false},
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
        PackedIndex ix = pack_records(recs);
        // the FASTA this was packed from: the other tools take their reference bases from <prefix>.vsc instead of parsing the
        // FASTA again as long as size and modification time still match (vsc_host.hpp, open_packed_for)
        (void)file_stamp(genome, &ix.src_size, &ix.src_mtime);
        write_index(prefix, ix);
        // a seed index file of an earlier genome under this prefix must not outlive the planes it was built from
        // (bidir_mapping loads <prefix>.vsi whenever it exists; -S below writes the new one)
        std::remove(seed_index_path(prefix).c_str());
        if (opts[2].set) {
            vsc_ctx *ctx = nullptr;
            vsc_genome *genome = nullptr;
            int st = vsc_ctx_create(0, &ctx);
            if (st != VSC_OK) throw std::runtime_error(st == VSC_ERR_NODEVICE ? "no HIP device available (-S needs one)" : "could not create the device context");
            st = vsc_genome_load(ctx, ix.hi.data(), ix.lo.data(), ix.nm.data(), 0, ix.hi.size(), ix.hi.size(), ix.contigs.data(),
                                 (uint32_t)ix.contigs.size(), &genome);
            if (st == VSC_OK) st = vsc_genome_build_index(ctx, genome, nullptr);
            if (st == VSC_OK) st = vsc_genome_index_save(ctx, genome, seed_index_path(prefix).c_str());
            const std::string why = st == VSC_OK ? "" : vsc_last_error(ctx);
            if (genome) vsc_genome_free(genome);
            vsc_ctx_destroy(ctx);
            if (st != VSC_OK) throw std::runtime_error(why);
        }
        std::printf("Index created successfully\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
