// vcf_loader - drop-in for VARSCOT_pipeline/variant_processing/vcf_loader.cpp: same six positional
// arguments, same progress lines, writes the SNP-genome FASTA (every haplotype combination of nearby
// variants as one short contig, ids chr_start_REF / chr_start_ALT_pos_ref_alt...).
// Host C++; the windows are searched afterwards by bidir_mapping (no FM index is built for them).
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <fstream>
#include <iostream>

#include "merge_host.hpp"
#include "vcf_expand.hpp"
#include "vsc_host.hpp"

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
    if (argc != 7) {
        std::cerr << "USAGE: vcf_loader FILE.vcf SNPGENOME.fa GENOME.fa SAMPLE SEQLENGTH THREADS\n";
        return 1;
    }
    unsigned sample = 0, seq_len = 0, threads = 0;
    const char *names[3] = {argv[4], argv[5], argv[6]};
    unsigned *vals[3] = {&sample, &seq_len, &threads};
    for (int i = 0; i < 3; ++i)
        if (!to_unsigned(names[i], vals[i])) {
            std::cerr << "ERROR: Cannot cast " << names[i] << " into an unsigned.\n";
            return 1;
        }
    try {
        std::cout << "Process records" << std::endl;
        std::ifstream vcf(argv[1]);
        if (!vcf) throw std::runtime_error("ERROR: Could not open VCF file.");
        auto chrs = vsc_vcf::read_vcf(vcf, sample);

        std::cout << "Compute overlap sequences" << std::endl;
        // reference bases, addressed by the first word of each id (FAI rule; region clamped as write_fasta.h:255-260): from
        // the packed genome that stands for the FASTA when there is one, else from the FASTA text in memory
        const vsc_merge::Genome genome(argv[3]);
        auto fetch = [&](const std::string &chr, uint32_t b, uint32_t e) -> std::string { return genome.region(chr, b, e, '+'); };

        std::cout << "Write fasta" << std::endl;
        std::ofstream out(argv[2]);
        if (!out) throw std::runtime_error("ERROR: Could not open single FASTA output file.");
        auto emit = [&](const std::string &id, const std::string &seq) {
            out << '>' << id << '\n';
            for (size_t i = 0; i < seq.size(); i += 70) out << seq.substr(i, 70) << '\n';  // SeqAn wraps at 70
            if (seq.empty()) out << '\n';
        };
        vsc_vcf::expand(chrs, seq_len, fetch, emit);
    } catch (const std::exception &e) {
        std::cout << e.what() << std::endl;
        return 1;
    }
    return 0;
}
