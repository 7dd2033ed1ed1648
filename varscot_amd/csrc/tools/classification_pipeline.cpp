// classification_pipeline - drop-in for `Rscript classificationPipeline.R OUT.txt FEATURES.txt TRUE|FALSE`
// (VARSCOT_pipeline/classification/classificationPipeline.R:21-51): reads the feature matrix the
// merger wrote, predicts every row with the reference's trained random forest on the GPU
// (vsc_rf_predict) and overwrites the Score column of the TSV with the probability of class "1"
// (TRUE) or the class (FALSE), as write.table(quote=FALSE, sep="\t", row.names=FALSE) formats it.
// Model: $VARSCOT_RF_MODEL or <dir of this tool>/../models/rfClassifier.vscrf (the forest of
// classification/rfClassifier.RData, exported by tests/golden/export_rf_model.py).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <unistd.h>

#include "forest_host.hpp"
#include "merge_host.hpp"

using vsc_forest::Forest;

static std::vector<std::string> split_tabs(const std::string &line)
{
    std::vector<std::string> f;
    size_t b = 0;
    for (;;) {
        size_t e = line.find('\t', b);
        f.push_back(line.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return f;
}

int main(int argc, char **argv)
{
    if (argc != 4) {
        std::cerr << "Error: Wrong number of arguments supplied. Arguments must be offtarget path (pipeline output), "
                     "feature matrix path and probability (true/false).\n";
        return 1;
    }
    try {
        const bool prob = std::string(argv[3]) == "TRUE" || std::string(argv[3]) == "true" || std::string(argv[3]) == "T";
        Forest forest = vsc_forest::load_forest(vsc_forest::default_model_path(argv[0]));

        // feature matrix: header with 443 names, then row name + 442 integers + activity
        std::ifstream fin(argv[2]);
        if (!fin) throw std::runtime_error(std::string("Error: cannot open file '") + argv[2] + "'");
        std::string line;
        std::getline(fin, line);
        const auto header = split_tabs(line);
        std::map<std::string, size_t> col;
        for (size_t i = 0; i < header.size(); ++i) col[header[i]] = i;
        if (header.size() != VSC_N_FEATURES + 1 || !col.count("ontargetActivity"))
            throw std::runtime_error("Error: the feature matrix does not have the 443 expected columns");
        vsc_forest::bind_features(forest, [&](const std::string &name) { return col.count(name) != 0; });
        // the rows: name, 442 small integers, activity.  Read in pieces that end at a line break, every piece parsed
        // by all host threads (a stream-and-split loop took 12 s for 2.6 M rows)
        std::vector<uint8_t> feat;
        std::vector<double> act;
        {
            const unsigned n_threads = vsc_host::host_threads();
            const size_t piece = (size_t)256 << 20;
            std::string buf, carry;
            std::vector<std::vector<uint8_t>> tf(n_threads);
            std::vector<std::vector<double>> ta(n_threads);
            std::vector<std::string> terr(n_threads);
            for (;;) {
                buf = carry;
                const size_t had = buf.size();
                buf.resize(had + piece);
                fin.read(&buf[had], (std::streamsize)piece);
                buf.resize(had + (size_t)fin.gcount());
                if (buf.empty()) break;
                const bool last = fin.eof();
                size_t end = buf.size();
                if (!last) {
                    const size_t nl = buf.rfind('\n');
                    if (nl == std::string::npos) throw std::runtime_error("Error: malformed feature matrix row");
                    end = nl + 1;
                }
                carry.assign(buf, end, std::string::npos);
                // line-aligned parts
                std::vector<size_t> cut(n_threads + 1, end);
                cut[0] = 0;
                for (unsigned t = 1; t < n_threads; ++t) {
                    size_t c = std::max(cut[t - 1], end * t / n_threads);
                    while (c < end && c > 0 && buf[c - 1] != '\n') ++c;
                    cut[t] = c;
                }
                auto parse = [&](unsigned t) {
                    auto &f = tf[t];
                    auto &a = ta[t];
                    f.clear();
                    a.clear();
                    const char *q = buf.data() + cut[t], *const stop = buf.data() + cut[t + 1];
                    while (q < stop) {
                        const char *eol = (const char *)std::memchr(q, '\n', (size_t)(stop - q));
                        if (!eol) eol = stop;
                        if (eol == q) { ++q; continue; }  // empty line
                        const char *c = (const char *)std::memchr(q, '\t', (size_t)(eol - q));  // past the row name
                        int k = 0;
                        while (c && k < VSC_N_FEATURES) {
                            ++c;
                            unsigned v = 0;
                            const char *d = c;
                            while (d < eol && *d >= '0' && *d <= '9') v = v * 10 + (unsigned)(*d++ - '0');
                            if (d == c || d >= eol || *d != '\t') { c = nullptr; break; }
                            f.push_back((uint8_t)v);
                            c = d;
                            ++k;
                        }
                        if (!c || k != VSC_N_FEATURES || std::memchr(c + 1, '\t', (size_t)(eol - c - 1))) {
                            terr[t] = "Error: malformed feature matrix row";
                            return;
                        }
                        a.push_back(std::strtod(std::string(c + 1, eol).c_str(), nullptr));
                        q = eol + 1;
                    }
                };
                std::vector<std::thread> workers;
                for (unsigned t = 0; t < n_threads; ++t) workers.emplace_back(parse, t);
                for (auto &w : workers) w.join();
                for (unsigned t = 0; t < n_threads; ++t) {
                    if (!terr[t].empty()) throw std::runtime_error(terr[t]);
                    feat.insert(feat.end(), tf[t].begin(), tf[t].end());
                    act.insert(act.end(), ta[t].begin(), ta[t].end());
                }
                if (last) break;
            }
        }
        const size_t n = act.size();

        // the TSV: read.table(header = FALSE) skips the '#' header line as a comment; only the Score column changes
        std::ifstream tin(argv[1]);
        if (!tin) throw std::runtime_error(std::string("Error: cannot open file '") + argv[1] + "'");
        std::vector<std::string> rows;
        while (std::getline(tin, line)) {
            if (line.empty() || line[0] == '#') continue;
            rows.push_back(line);
        }
        tin.close();
        if (rows.size() != n) throw std::runtime_error("Error: replacement has a different number of rows than the data");

        std::vector<double> p(n);
        std::vector<uint8_t> cls(n), tie(n);
        if (n) {
            vsc_ctx *ctx = nullptr;
            int st = vsc_ctx_create(0, &ctx);
            if (st != VSC_OK) throw std::runtime_error(st == VSC_ERR_NODEVICE ? "Error: no HIP device available (there is no CPU fallback)." : "Error: could not create the device context.");
            const vsc_rf_model m = vsc_forest::model_of(forest);
            st = vsc_rf_predict(ctx, &m, feat.data(), act.data(), n, p.data(), cls.data(), tie.data());
            const std::string err = st == VSC_OK ? "" : vsc_last_error(ctx);
            vsc_ctx_destroy(ctx);
            if (st != VSC_OK) throw std::runtime_error("Error: " + err);
        }
        std::ofstream out(argv[1]);
        if (!out) throw std::runtime_error(std::string("Error: cannot open file '") + argv[1] + "'");
        const size_t ncol = rows.empty() ? 9 : (size_t)std::count(rows[0].begin(), rows[0].end(), '\t') + 1;
        std::string text = "#Chr\tStart\tEnd\tName\tScore\tStrand\tSequence\tMismatch_Number\tMismatch_Positions";
        text += ncol == 10 ? "\tVariants\n" : "\n";
        char buf[64];
        for (size_t i = 0; i < n; ++i) {
            const std::string &r = rows[i];
            size_t b = 0;  // start of the fifth field
            int tabs = 0;
            while (tabs < 4 && (b = r.find('\t', b)) != std::string::npos) { ++b; ++tabs; }
            if (tabs < 4) continue;  // fewer than five fields
            size_t e = r.find('\t', b);
            if (e == std::string::npos) e = r.size();
            text.append(r, 0, b);
            if (prob) {
                std::snprintf(buf, sizeof buf, "%.15g", p[i]);
                text += buf;
            } else {
                text += cls[i] ? "1" : "0";  // an exact 500/500 vote (R: random) is reported as "0"
            }
            text.append(r, e, std::string::npos);
            text += '\n';
            if (text.size() > ((size_t)64 << 20)) {
                out.write(text.data(), (std::streamsize)text.size());
                text.clear();
            }
        }
        out.write(text.data(), (std::streamsize)text.size());
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
