/*
 * varscot_hip_debug.h - test and experiment hooks of libvarscot_hip.so.
 *
 * NOT part of the drop-in boundary (include/varscot_hip.h): nothing here is needed to use the library, and a
 * caller that never includes this header gets the defaults every measurement in DESIGN.md was taken with.  The
 * hooks exist so that tests can drive the rare paths of the library on inputs of a few thousand records (many
 * sort levels, several scoring passes, a failing RCCL load) and so that experiments can vary launch constants
 * without rebuilding.  They are explicit calls on a context: the library reads NO environment variable.
 */
#ifndef VARSCOT_HIP_DEBUG_H
#define VARSCOT_HIP_DEBUG_H

#include "varscot_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 (or -1 where stated) = the library's default. */
typedef struct {
    uint32_t seed_groups_per_cu; /* seed_sliced_kernel: resident workgroups per CU */
    uint32_t seed_reserve;       /* record slots a wave reserves per atomic (rounded to a power of two in 64..1024) */
    uint32_t sort_cap;           /* LDS capacity of the finalize kernel in records (16 .. 8064): small values force levels */
    uint32_t sort_max_bits;      /* key bits per partition level (1 .. 11) */
    int32_t sort_xcd;            /* -1 default; 0: partition tiles in index order; 1: one contiguous eighth per XCD */
    uint32_t sort_debug;         /* 1: per-level statistics on stderr; 2: every bin recounted on the host */
    int32_t sort_optimistic;     /* -1 default; 0: always histogram first; 1: slot partition first whenever a level partitions */
    uint32_t sort_slot_cap;      /* record slots per bin of the slot partition (default: derived from sort_cap) */
    uint64_t score_chunk;        /* rows per scoring pass */
    int32_t score_slices;        /* -1 default; 0 / 1: slice-major scoring order off / forced */
    uint32_t score_slice_shift;  /* log2 positions per slice (8 .. 31) */
    int32_t seed_shared;         /* -1 default; 0 / 1: one chunk per wave / one chunk per workgroup in seed_sliced_kernel */
    int32_t seed_group_out;      /* -1 default (on with chunk sharing); 0 / 1: per-wave / workgroup-shared open output blocks */
    int32_t seed_tight;          /* -1 default: segment thresholds follow what the site's PAM leaves of the limit, the cut chosen by the
                                    cost model; 0: floor(m / 3) in all three; 1 + k0 + 3 k1: segments 0 / 1 within k0 / k1 substitutions (if that is a valid cut) */
    int32_t rf_form;             /* -1 default (the best node form the forest allows); 0: plain nodes; 1: at most the compact form; 2 = -1 */
    uint32_t reserved[1];
} vsc_debug_params;

/* Replaces the context's hooks (NULL: back to the defaults). */
int vsc_ctx_set_debug_params(vsc_ctx *ctx, const vsc_debug_params *params);
int vsc_ctx_get_debug_params(const vsc_ctx *ctx, vsc_debug_params *out);

/* A context whose stream may only use the compute units of `cu_mask` (bit i of word i / 32 = CU i, n_words words:
 * hipExtStreamCreateWithCUMask) - experiments with kernels of two contexts on disjoint parts of the device (tools/cu_mask_probe.py). */
int vsc_ctx_create_masked(int device_id, const uint32_t *cu_mask, uint32_t n_words, vsc_ctx **out);

/* Host-side lap times of vsc_search and vsc_windows_build on stderr (process-wide; 0 = off). */
void vsc_debug_set_host_timing(int on);

typedef struct {
    int32_t rccl;              /* -1: RCCL when n > 1 distinct devices; 0: device copies; 1: insist on RCCL (also n = 1);
                                  2: try RCCL also for n = 1, device copies when it cannot be set up */
    const char *rccl_library;  /* NULL: librccl.so.1 / librccl.so; else the one name to dlopen (tests: a name that fails) */
} vsc_multi_debug_params;
/* vsc_multi_create with hooks (NULL = vsc_multi_create). */
int vsc_multi_create_debug(const int *device_ids, int n, const vsc_multi_debug_params *params, vsc_multi **out);

#ifdef __cplusplus
}
#endif
#endif
