// fasta_writer - drop-in for VARSCOT_pipeline/variant_processing/fasta_writer.cpp
// (writeFastaOntargets, extract_fasta_ontargets.h:92-139): BED6 on-targets -> OUTPUT1 (the 23-mers the
// mapper searches) and OUTPUT2 (the 30-mers with flanks for the on-target activity model:
// 4 upstream + 3 downstream, strand-aware, :44-53).  Sequences of '-' records are reverse-complemented.
#include <fstream>
#include <iostream>
#include <sstream>

#include "merge_host.hpp"

int main(int argc, char **argv)
{
    if (argc != 5) {
        std::cerr << "USAGE: extract_fasta_ontargets OUTPUT1.fa OUTPUT2.fa ONTARGETS.bed GENOME.fa\n";
        return 1;
    }
    try {
        const vsc_merge::Genome genome(argv[4]);
        for (int flanking = 0; flanking < 2; ++flanking) {
            std::ofstream out(argv[1 + flanking]);
            if (!out) throw std::runtime_error("ERROR: Could not open output file.");
            std::ifstream bed(argv[3]);
            if (!bed) throw std::runtime_error("ERROR: Could not open BED file.");
            std::string line;
            while (std::getline(bed, line)) {
                if (line.empty() || line[0] == '#') continue;
                std::istringstream is(line);
                std::string chr, name, score, strand;
                unsigned long start = 0, end = 0;
                if (!(is >> chr >> start >> end >> name >> score >> strand)) continue;
                const char s = strand.empty() ? '+' : strand[0];
                uint32_t b = (uint32_t)start, e = (uint32_t)end;
                if (flanking && s == '+') b -= 4, e += 3;  // unsigned arithmetic as in the reference (:46-52)
                if (flanking && s == '-') b -= 3, e += 4;
                const std::string seq = genome.region(chr, b, e, s);
                out << '>' << name << '\n';
                for (size_t i = 0; i < seq.size(); i += 70) out << seq.substr(i, 70) << '\n';
                if (seq.empty()) out << '\n';
            }
        }
    } catch (const std::exception &e) {
        std::cout << e.what() << std::endl;
        return 1;
    }
    return 0;
}
