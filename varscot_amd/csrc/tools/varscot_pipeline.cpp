// varscot_pipeline - the stages of the VARSCOT driver in ONE process, for runs that do not need the intermediate files:
// on-targets from the packed genome -> reference search on the GPU -> [per VCF sample: alt-allele windows straight from
// the packed planes (vsc_windows_build), searched as a second genome] -> merge / filters -> scores (MIT, or feature
// matrix [+ the trained forest]) -> TSV sorted on the name column + feature matrix.  Same rows, same text as
//   fasta_writer | bidir_index | bidir_mapping | vcf_loader | bidir_index | bidir_mapping | bam_merger[_ref_only] |
//   classification_pipeline | sort
// (VARSCOT_pipeline/VARSCOT:260-357) - tests/test_pipeline.py holds the two routes byte-equal - without the SAM text
// (640 MB for 1 000 reads at 6 mismatches), the SNP-genome FASTA (8.9 M records for 5 M SNPs), a second pass over the
// genome FASTA per stage, or the feature matrix being parsed back for the classifier.
// What each step replaces: extract_fasta_ontargets.h:33-139 (on-targets), read_mapping/bidir_mapping.cpp:285-309 (search +
// SAM order), variant_processing/vcf_loader.cpp:40-68 (windows), filter_output_bam.h:70-124,279-317,362-418 (records ->
// potential off-targets, filters), merge_output_bam.h:46-720 (TSV / feature matrix), classification/
// classificationPipeline.R:21-49 (forest).  Host C++; every computation on hits runs in libvarscot_hip.so.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>

#include "forest_host.hpp"
#include "merge_host.hpp"

using namespace vsc_host;
using namespace vsc_merge;

namespace {

struct Lap {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = false;
    void operator()(const char *what)
    {
        const auto t1 = std::chrono::steady_clock::now();
        if (on) std::fprintf(stderr, "[varscot_pipeline] %-34s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

// The records of one search in the order bidir_mapping writes them (vsc_sam_order), as the potential off-targets
// readBamFile makes of the SAM lines (filter_output_bam.h:362-418): target = read name, chr = contig name, sequence =
// the window (reverse-complemented for '-'), mismatch positions = what getMismatchPositions recovers from the MD string.
// name(contig) / window(contig, pos) give the contig's id and the 23 forward bases.
template <class Name, class Window>
std::vector<OffTarget> rows_of_hits(vsc_hits *hits, vsc_ctx *ctx, const std::vector<std::string> &read_names, int md_style, Name &&name,
                                    Window &&window)
{
    const uint64_t n = vsc_hits_count(hits);
    const vsc_hit *h = nullptr;
    if (vsc_hits_data(hits, &h) != VSC_OK) throw std::runtime_error(std::string("ERROR: ") + vsc_last_error(ctx));
    std::vector<uint64_t> order(n);
    std::vector<uint8_t> secondary(n);
    vsc_sam_order(h, n, order.data(), secondary.data());
    std::vector<OffTarget> out(n);
    const unsigned n_threads = host_threads();
    std::vector<std::thread> pool;
    std::vector<std::string> errors(n_threads);
    for (unsigned t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t] {
            try {
                for (uint64_t i = n * t / n_threads; i < n * (t + 1) / n_threads; ++i) {
                    const vsc_hit &r = h[order[i]];
                    OffTarget &p = out[i];
                    p.target = read_names[r.guide];
                    p.chr = name(r.contig);
                    p.pos = r.pos;
                    p.strand = VSC_HIT_STRAND(r.info) ? '-' : '+';
                    const std::string fwd = window(r.contig, r.pos);
                    p.mm = md_positions(md_string(VSC_HIT_MASK(r.info), fwd.c_str(), md_style));
                    p.sequence = fwd;
                    if (p.strand == '-') revcomp_in_place(p.sequence);
                    p.snp_type = "REF";
                }
            } catch (const std::exception &e) {
                errors[t] = e.what();
            }
        });
    for (auto &th : pool) th.join();
    for (const auto &e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    return out;
}

std::vector<unsigned> parse_list(const std::string &s)
{
    std::vector<unsigned> out;
    size_t b = 0;
    for (;;) {
        const size_t e = s.find(',', b);
        const std::string item = s.substr(b, e == std::string::npos ? std::string::npos : e - b);
        char *end = nullptr;
        const unsigned long v = std::strtoul(item.c_str(), &end, 10);
        if (item.empty() || *end) throw std::runtime_error("bad list '" + s + "'");
        out.push_back((unsigned)v);
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}

}  // namespace

int main(int argc, char **argv)
{
    std::vector<Option> opts = {
        {'b', "bed", "On-targets (BED6, 23 bp)", true},
        {'g', "genome", "Genome FASTA (only read when the packed genome is missing or stale)", true},
        {'i', "index", "Prefix of the packed genome (<prefix>.vsc; written if missing)", true},
        {'o', "output-stem", "Result path without its .txt: STEM.txt + STEM_feature_matrix.txt (several samples: STEM_sample<k>...)", true},
        {'a', "activity", "On-target activity table (ID Sequence Score ...)", true},
        {'m', "mismatches", "Maximum number of mismatches (0..8)", true},
        {'p', "pam", "Additional non-canonical PAM", false},
        {'f', "vcf", "Variants (VCF)", false},
        {'s', "samples", "VCF sample columns, comma-separated (default 0)", false},
        {'e', "evaluation", "mit (default) | class | prob", false},
        {'S', "md-style", "0 = SAM-spec MD strings (default), 1 = no zeros between adjacent mismatches", false},
        {'D', "device", "HIP device (default 0), or a comma-separated list (0,1,2,...): the reference genome is sharded over those devices for its "
                        "search (vsc_multi_search: one gather to the first), everything else runs on the first", false},
        {'t', "threads", "Host threads for VCF parsing and text", false},
        {'V', "verbose", "Stage times on stderr", false, false},
    };
    const int pr = parse_args(argc, argv, opts, "VARSCOT - one-process pipeline",
                              "On-targets -> off-target search (reference and variant windows) -> merge -> scores -> TSV, without "
                              "intermediate files. Same result files as the staged tools.");
    if (pr) return pr == 1;
    auto val = [&](char c) -> const Option & {
        for (const auto &o : opts)
            if (o.short_name == c) return o;
        throw std::logic_error("option");
    };
    Lap lap;
    lap.on = val('V').set;
    vsc_ctx *ctx = nullptr;
    vsc_genome *genome = nullptr;
    vsc_multi *multi = nullptr;
    int rc = 1;
    try {
        char *end = nullptr;
        const long mm = std::strtol(val('m').value.c_str(), &end, 10);
        if (end == val('m').value.c_str() || *end || mm < 0 || mm > 8) throw std::runtime_error("Error: Maximum number of mismatches must lie between 0 and 8.");
        const std::string evaluation = val('e').set ? val('e').value : "mit";
        if (evaluation != "mit" && evaluation != "class" && evaluation != "prob") throw std::runtime_error("Error: -e must be mit, class or prob.");
        const bool features = evaluation != "mit";
        const int md_style = val('S').set ? std::atoi(val('S').value.c_str()) : 0;
        std::vector<int> devices;
        for (unsigned d : parse_list(val('D').set ? val('D').value : "0")) devices.push_back((int)d);
        const int device = devices[0];
        const unsigned threads = val('t').set ? (unsigned)std::atoi(val('t').value.c_str()) : 0;
        const std::string stem = val('o').value, prefix = val('i').value, fasta = val('g').value;
        const std::string pam = val('p').set ? val('p').value : "";
        std::vector<unsigned> samples;
        if (val('f').set) samples = parse_list(val('s').set ? val('s').value : "0");

        // ---- the packed reference: <prefix>.vsc, built from the FASTA (what `bidir_index` does) when missing -------
        if (!std::ifstream(index_path(prefix)).good()) {
            const auto recs = read_fasta(fasta);
            PackedIndex ix = pack_records(recs);
            (void)file_stamp(fasta, &ix.src_size, &ix.src_mtime);
            write_index(prefix, ix);
            std::remove(seed_index_path(prefix).c_str());
            lap("packed the genome FASTA");
        }
        setenv("VARSCOT_PACKED_GENOME", prefix.c_str(), 1);
        const Genome ref(fasta);  // (a stale <prefix>.vsc falls back to the FASTA text)
        PackedView packed_own;    // the planes to upload: the Genome's mapping, or the file's whatever it was packed from
        const PackedView *planes = &ref.packed;
        if (!ref.from_packed) {
            if (!packed_own.open(index_path(prefix))) throw std::runtime_error("Could not open index " + index_path(prefix));
            planes = &packed_own;
        }
        lap("opened the packed genome");

        // ---- on-targets: the reads are the 23-mers of the BED records (fasta_writer's first output) -----------------
        std::map<std::string, OffTarget> on;
        std::map<std::string, unsigned> count;
        read_ontargets(val('b').value, ref, on, count);
        std::vector<std::string> read_names, read_seqs;
        {
            std::ifstream bed(val('b').value);
            std::string line;
            while (std::getline(bed, line)) {  // BED order, every record (also a repeated name), as writeFastaOntargets walks them
                if (line.empty() || line[0] == '#') continue;
                std::istringstream is(line);
                std::string chr, name, score, strand;
                unsigned long start = 0, stop = 0;
                if (!(is >> chr >> start >> stop >> name >> score >> strand)) continue;
                read_names.push_back(name);
                read_seqs.push_back(ref.region(chr, (uint32_t)start, (uint32_t)stop, strand.empty() ? '+' : strand[0]));
            }
        }
        std::vector<uint64_t> codes(read_seqs.size());
        for (size_t i = 0; i < read_seqs.size(); ++i) {
            if (read_seqs[i].size() != VSC_READ_LEN)
                throw std::runtime_error("read '" + read_names[i] + "' is not 23 nt long (VARSCOT searches 20 nt + PAM)");
            codes[i] = vsc_pack_guide(read_seqs[i].c_str());
        }
        const auto activity = read_tuscan(val('a').value);
        vsc_forest::Forest forest;
        vsc_rf_model model{};
        if (features) {
            forest = vsc_forest::load_forest(vsc_forest::default_model_path(argv[0]));
            vsc_forest::bind_features(forest, [](const std::string &) { return true; });
            model = vsc_forest::model_of(forest);
        }
        lap("on-targets, activity, forest");

        // ---- reference search ------------------------------------------------------------------------------------
        int st = vsc_ctx_create(device, &ctx);
        if (st != VSC_OK) throw std::runtime_error(st == VSC_ERR_NODEVICE ? "no HIP device available (there is no CPU fallback)" : "could not create the device context");
        st = vsc_genome_load(ctx, planes->hi, planes->lo, planes->nm, 0, planes->n_words, planes->n_words, planes->contigs.data(),
                             (uint32_t)planes->contigs.size(), &genome);
        if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
        if (pam.size() != 2 && std::ifstream(seed_index_path(prefix)).good() &&
            vsc_genome_index_load(ctx, genome, seed_index_path(prefix).c_str()) != VSC_OK)
            std::fprintf(stderr, "%s: %s - building the seed index instead\n", argv[0], vsc_last_error(ctx));
        lap("genome to the device");
        vsc_search_params sp{};
        sp.max_mismatches = (uint32_t)mm;
        if (pam.size() == 2) {
            sp.has_extra_pam = 1;
            sp.extra_pam[0] = pam[0];
            sp.extra_pam[1] = pam[1];
        }
        vsc_hits *hits = nullptr;
        if (devices.size() > 1) {
            // the reference genome over several devices (what `bidir_mapping -D 0,1,...` does): shards, one gather, merge
            vsc_genome_free(genome);
            genome = nullptr;
            vsc_multi_genome *mg = nullptr;
            st = vsc_multi_create(devices.data(), (int)devices.size(), &multi);
            if (st != VSC_OK) throw std::runtime_error("could not create the device contexts");
            st = vsc_multi_genome_load(multi, planes->hi, planes->lo, planes->nm, planes->n_words, planes->contigs.data(),
                                       (uint32_t)planes->contigs.size(), &mg);
            if (st == VSC_OK) st = vsc_multi_search(multi, mg, codes.data(), (uint32_t)codes.size(), &sp, &hits);
            const std::string why = st == VSC_OK ? "" : vsc_multi_last_error(multi);
            vsc_multi_genome_free(mg);
            if (st != VSC_OK) throw std::runtime_error(why);
        } else {
            st = vsc_search(ctx, genome, codes.data(), (uint32_t)codes.size(), &sp, &hits);
            if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
        }
        lap("reference search");
        const std::vector<OffTarget> ref_hits = rows_of_hits(
            hits, ctx, read_names, md_style, [&](uint32_t c) -> const std::string & { return planes->names[c]; },
            [&](uint32_t c, uint32_t pos) { return planes->bases(c, pos, VSC_READ_LEN); });
        vsc_hits_free(hits);
        if (genome) vsc_genome_free(genome);
        genome = nullptr;
        if (multi) vsc_multi_destroy(multi);
        multi = nullptr;
        lap("reference records -> off-targets");

        OutputOptions oo;
        oo.ctx = ctx;
        oo.forest = features ? &model : nullptr;
        oo.prob = evaluation == "prob";
        oo.sort_by_name = true;
        const unsigned seq_len = VSC_READ_LEN;
        if (samples.empty()) {
            // processRefOnly, merge_output_bam.h:485-720
            std::vector<const OffTarget *> rows;
            for (const auto &h : ref_hits)
                if (!same(h, on.at(h.target))) rows.push_back(&h);
            const std::string fpath = stem + "_feature_matrix.txt";
            write_outputs(stem + ".txt", features ? &fpath : nullptr, false, rows, on, count, activity, device, oo);
            lap("scores + text");
        }
        std::vector<const char *> cnames;
        for (const auto &n : planes->names) cnames.push_back(n.c_str());
        for (unsigned sample : samples) {
            // ---- the sample's alt-allele windows, straight from the packed planes (vcf_loader + bidir_index on the SNP genome)
            vsc_windows *win = nullptr;
            char err[512] = {0};
            st = vsc_windows_build(val('f').value.c_str(), sample, seq_len, threads, planes->hi, planes->lo, planes->nm, planes->contigs.data(),
                                   cnames.data(), (uint32_t)cnames.size(), &win, err, sizeof err);
            if (st != VSC_OK) throw std::runtime_error(err[0] ? err : "ERROR: could not build the variant windows.");
            const uint32_t n_win = vsc_windows_count(win);
            if (n_win == 0) {
                vsc_windows_free(win);
                throw std::runtime_error("Error: SNP genome is empty");
            }
            lap("variant windows");
            const vsc_contig *wtab = vsc_windows_contigs(win);
            const uint32_t *whi = vsc_windows_plane(win, 0), *wlo = vsc_windows_plane(win, 1), *wnm = vsc_windows_plane(win, 2);
            const uint64_t wwords = vsc_windows_words(win);
            vsc_genome *snp = nullptr;
            st = vsc_genome_load(ctx, whi, wlo, wnm, 0, wwords, wwords, wtab, n_win, &snp);
            if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
            vsc_hits *shits = nullptr;
            st = vsc_search(ctx, snp, codes.data(), (uint32_t)codes.size(), &sp, &shits);
            if (st != VSC_OK) throw std::runtime_error(vsc_last_error(ctx));
            lap("window search");
            auto wname = [&](uint32_t c) {
                uint32_t len = 0;
                const char *p = vsc_windows_name(win, c, &len);
                return std::string(p, len);
            };
            std::vector<OffTarget> snp_hits = rows_of_hits(shits, ctx, read_names, md_style, wname, [&](uint32_t c, uint32_t pos) {
                std::string s(VSC_READ_LEN, 'N');
                vsc_unpack_bases(whi, wlo, wnm, wtab[c].offset + pos, VSC_READ_LEN, &s[0]);
                return s;
            });
            vsc_hits_free(shits);
            vsc_genome_free(snp);
            // mergeResults, merge_output_bam.h:46-460: reference hits that are not the on-target and not shadowed by a window
            // (filterRefAlignment, filter_output_bam.h:70-124), then the window hits with their coordinates restored
            // (filterSnpAlignment, :279-317)
            const WindowIndex windows(
                n_win,
                [&](size_t i, size_t *len) {
                    uint32_t l = 0;
                    const char *p = vsc_windows_name(win, (uint32_t)i, &l);
                    *len = l;
                    return p;
                },
                [&](size_t i) { return (uint64_t)wtab[i].length; });
            lap("window records, shadow table");
            std::map<std::string, unsigned> cnt = count;  // every sample numbers its rows from 1
            std::vector<const OffTarget *> rows;
            for (const auto &h : ref_hits)
                if (!same(h, on.at(h.target)) && !windows.shadows(h.chr, h.pos, seq_len)) rows.push_back(&h);
            for (size_t i = 0; i < snp_hits.size(); ++i) {
                OffTarget &h = snp_hits[i];
                const auto id = split_id(h.chr);
                h.chr = id[0];
                h.pos = h.pos + (uint32_t)c_atoi(id.size() > 1 ? id[1] : "0");
                snp_type(h.snp_type, id, h.pos, seq_len);
                bool valid = !same(h, on.at(h.target));
                if (i > 0 && same(h, snp_hits[i - 1])) valid = false;
                if (valid) rows.push_back(&h);
            }
            const std::string tag = samples.size() > 1 ? "_sample" + std::to_string(sample) : "";
            const std::string fpath = stem + tag + "_feature_matrix.txt";
            write_outputs(stem + tag + ".txt", features ? &fpath : nullptr, true, rows, on, cnt, activity, device, oo);
            vsc_windows_free(win);
            lap("merge, scores + text");
        }
        rc = 0;
    } catch (const std::exception &e) {
        std::cout << e.what() << std::endl;
        rc = 1;
    }
    if (genome) vsc_genome_free(genome);
    if (multi) vsc_multi_destroy(multi);
    if (ctx) vsc_ctx_destroy(ctx);
    return rc;
}
