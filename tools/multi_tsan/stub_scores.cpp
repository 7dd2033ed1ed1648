// stub_scores.cpp - the five device entry points the mergers (tools/bam_merger*.cpp, tools/merge_host.hpp) call, answered on
// the host with INVENTED values, so that the mergers' own code - SAM / BED / activity parsing, the packed-genome and FASTA
// back ends, on-target removal, the shadow filter, getSnpType's coordinate restoration, the text formatting on all host
// threads - can run under AddressSanitizer / UBSan / ThreadSanitizer on a machine without a GPU (tools/multi_tsan/run.sh).
// TEST INFRASTRUCTURE ONLY: the scores are not the reference's (MIT = 1 / (1 + mismatches), all-zero feature rows with the
// mismatch count in column 0), the product links libvarscot_hip.so and has no CPU path.
#include <string>

#include "stub_host.h"
#include "varscot_hip.h"

extern "C" {
int vsc_ctx_create(int, vsc_ctx **out)
{
    *out = new vsc_ctx();
    return VSC_OK;
}
int vsc_ctx_destroy(vsc_ctx *c)
{
    delete c;
    return VSC_OK;
}
const char *vsc_last_error(const vsc_ctx *c) { return c ? c->err.c_str() : "null context"; }
int vsc_score_pairs(vsc_ctx *, const uint64_t *, const uint64_t *, const uint32_t *masks, uint64_t n, double *mit, uint8_t *mit_flags, uint8_t *features)
{
    for (uint64_t i = 0; i < n; ++i) {
        const unsigned nm = (unsigned)__builtin_popcount(masks[i]);
        if (mit) mit[i] = 1.0 / (1.0 + nm);
        if (mit_flags) mit_flags[i] = 0;
        if (features) {
            for (int k = 0; k < VSC_N_FEATURES; ++k) features[i * VSC_N_FEATURES + k] = 0;
            features[i * VSC_N_FEATURES] = (uint8_t)nm;
        }
    }
    return VSC_OK;
}
int vsc_rf_predict(vsc_ctx *, const vsc_rf_model *, const uint8_t *features, const double *, uint64_t n, double *prob, uint8_t *cls, uint8_t *tie)
{
    for (uint64_t i = 0; i < n; ++i) {
        if (prob) prob[i] = features[i * VSC_N_FEATURES] < 3 ? 0.75 : 0.25;
        if (cls) cls[i] = features[i * VSC_N_FEATURES] < 3;
        if (tie) tie[i] = 0;
    }
    return VSC_OK;
}
}
