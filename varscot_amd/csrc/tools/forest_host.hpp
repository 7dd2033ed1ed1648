// forest_host.hpp - the trained random forest of the reference (classification/rfClassifier.RData, exported to
// models/rfClassifier.vscrf by tests/golden/export_rf_model.py) as the host tools load it and hand it to
// vsc_rf_predict: what classification/classificationPipeline.R:21-25 does with load() + the feature matrix's column names.
#pragma once

#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <unistd.h>

#include "merge_host.hpp"

namespace vsc_forest {

struct Forest {
    uint32_t n_trees = 0, n_nodes = 0;
    std::vector<std::string> names;
    std::vector<int8_t> status;
    std::vector<uint8_t> best_var, node_class;
    std::vector<uint16_t> left, right, feature;
    std::vector<double> split;
};

inline Forest load_forest(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("Error: could not open the classifier " + path);
    char magic[8];
    uint32_t hdr[3];
    in.read(magic, 8);
    in.read((char *)hdr, 12);
    if (!in || std::memcmp(magic, "VSCRF001", 8) != 0) throw std::runtime_error("Error: " + path + " is not a forest file");
    Forest f;
    f.n_trees = hdr[0];
    f.n_nodes = hdr[1];
    f.names.resize(hdr[2]);
    for (auto &n : f.names) {
        uint16_t l = 0;
        in.read((char *)&l, 2);
        n.resize(l);
        in.read(&n[0], l);
    }
    const size_t n = (size_t)f.n_trees * f.n_nodes;
    f.status.resize(n), f.best_var.resize(n), f.left.resize(n), f.right.resize(n), f.split.resize(n), f.node_class.resize(n);
    in.read((char *)f.status.data(), n);
    in.read((char *)f.best_var.data(), n);
    in.read((char *)f.left.data(), 2 * n);
    in.read((char *)f.right.data(), 2 * n);
    in.read((char *)f.split.data(), 8 * n);
    in.read((char *)f.node_class.data(), n);
    if (!in) throw std::runtime_error("Error: truncated forest file " + path);
    return f;
}

// $VARSCOT_RF_MODEL, or models/rfClassifier.vscrf next to the directory of the running tool
inline std::string default_model_path(const char *argv0)
{
    const char *env = std::getenv("VARSCOT_RF_MODEL");
    if (env && *env) return env;
    char exe[4096];
    const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
    std::string dir = n > 0 ? std::string(exe, (size_t)n) : std::string(argv0);
    dir = dir.substr(0, dir.find_last_of('/'));
    return dir + "/../models/rfClassifier.vscrf";
}

// forest.feature[node] = column of the dense feature row the node tests: the forest's variable names looked up among
// the 443 column names of the feature matrix (present(name): is the column in the file that is being classified?)
template <class Present> void bind_features(Forest &forest, Present &&present)
{
    const auto names = vsc_merge::feature_names();
    forest.feature.assign(forest.status.size(), 0);
    std::vector<uint16_t> col_of_var(forest.names.size());
    for (size_t v = 0; v < forest.names.size(); ++v) {
        size_t c = 0;
        while (c < names.size() && names[c] != forest.names[v]) ++c;
        if (c == names.size() || !present(forest.names[v])) throw std::runtime_error("Error: variables in the training data missing in newdata");
        col_of_var[v] = (uint16_t)c;
    }
    for (size_t i = 0; i < forest.status.size(); ++i)
        if (forest.best_var[i]) forest.feature[i] = col_of_var[forest.best_var[i] - 1];
}

inline vsc_rf_model model_of(const Forest &f)
{
    return vsc_rf_model{f.n_trees, f.n_nodes, f.status.data(), f.feature.data(), f.left.data(), f.right.data(), f.split.data(),
                        f.node_class.data()};
}

}  // namespace vsc_forest
