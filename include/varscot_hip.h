/*
 * varscot_hip.h - C ABI of libvarscot_hip.so, the MI355X (gfx950) implementation of VARSCOT's
 * genome-wide off-target search hot path.
 *
 * The reference (BauerLab/VARSCOT) has no plugin / FFI interface: its only seams are the argv +
 * file boundaries between stage binaries.  This ABI is therefore the boundary a maintainer would
 * bind *inside* those binaries; every entry point cites the reference code it replaces
 * (paths relative to VARSCOT_pipeline/).  INTEGRATION.md shows the binding.
 *
 * Conventions: every call returns an int status (0 = VSC_OK, negative errno-style otherwise); no
 * C++ exception crosses the boundary; no global state; a vsc_ctx is single-threaded (a process
 * may own several, one per device / stream); all sizes are explicit; all buffers little-endian;
 * pointers are HOST pointers unless the parameter name ends in _dev.
 *
 * Genome layout ("packed planes"): the genome is one global coordinate space.  Contig c occupies
 * global positions [offset_c, offset_c + length_c); consecutive contigs are separated by at least
 * one N position, and every position outside a contig is N.  Three bit planes describe it, one bit
 * per base, bit b of 32-bit word w <-> global position 32*w + b:
 *     hi, lo : the 2-bit base code (A = 00, C = 01, G = 10, T = 11 as hi:lo; complement = NOT both)
 *     nmask  : 1 where the position is N / a separator / padding (hi and lo are then ignored)
 * = 0.375 byte per base.  The total padded length must be < 2^32 - 4096 (the reference's own limit
 * is 4 Gbases: read_mapping/bidir_index.cpp:17, read_mapping/common.h:12-18).
 */
#ifndef VARSCOT_HIP_H
#define VARSCOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSC_ABI_VERSION 5

#define VSC_OK 0
#define VSC_ERR_INVALID (-22)  /* EINVAL: bad argument (e.g. mismatches outside 0..8)          */
#define VSC_ERR_NOMEM (-12)    /* ENOMEM: host or device allocation failed                      */
#define VSC_ERR_DEVICE (-5)    /* EIO:    a HIP call failed, see vsc_last_error()               */
#define VSC_ERR_RANGE (-34)    /* ERANGE: genome / result does not fit the 32-bit position space */
#define VSC_ERR_NODEVICE (-19) /* ENODEV: no usable gfx950 device                               */

#define VSC_READ_LEN 23 /* VARSCOT fixes SEQLENGTH=23 (VARSCOT:253; CIGAR "23M" bidir_mapping.cpp:102) */
#define VSC_MAX_MISMATCHES 8 /* read_mapping/bidir_mapping.cpp:234-238 */

typedef struct vsc_ctx vsc_ctx;
typedef struct vsc_genome vsc_genome;
typedef struct vsc_hits vsc_hits;

/* One contig of the global coordinate space. */
typedef struct {
    uint64_t offset; /* global position of the contig's first base */
    uint32_t length; /* number of bases */
    uint32_t reserved;
} vsc_contig;

/*
 * One candidate off-target site = one record the reference inserts into its `records` map
 * (read_mapping/bidir_mapping.cpp:88-125).
 *   info: bit 31       strand, 1 = '-' (BAM_FLAG_RC, :97-98)
 *         bits 23..27  NM = mismatches over all 23 positions (:79-86,121)
 *         bits 0..22   mismatch mask in forward-genome window coordinates, bit i = window position
 *                      i differs from fullRead[i] - the information the reference routes through
 *                      the MD tag (:114-122) and re-parses in
 *                      variant_processing/filter_output_bam.h:330-349.
 */
typedef struct {
    uint32_t guide;  /* index of the read in input order (:287) */
    uint32_t contig; /* record.r.rID (:99) */
    uint32_t pos;    /* record.r.beginPos, 0-based (:100) */
    uint32_t info;
} vsc_hit;

#define VSC_HIT_STRAND(info) (((info) >> 31) & 1u)
#define VSC_HIT_NM(info) (((info) >> 23) & 31u)
#define VSC_HIT_MASK(info) ((info)&0x7FFFFFu)

/* Search options = the -M and -P options of bidir_mapping (read_mapping/bidir_mapping.cpp:207-216). */
typedef struct {
    uint32_t max_mismatches; /* -M, 0..8 (:234-238) */
    uint8_t has_extra_pam;   /* -P given */
    char extra_pam[2];       /* -P: additional non-canonical PAM besides (N)GG and (N)GA (:240-247) */
    uint8_t algorithm;       /* VSC_ALGO_*: how the (identical) hit set is computed */
} vsc_search_params;

/* Both algorithms return the same records.  SCAN streams the packed planes and compares every
 * PAM-valid window with every read.  SEED keeps the PAM-valid windows filed by 7-base segments in HBM
 * (built once per genome and PAM set, vsc_genome_build_index) and compares a read only with the
 * windows that a pigeonhole argument cannot rule out - the role the FM-index halves play in
 * read_mapping/bidir_mapping.cpp:129-162.  AUTO picks SEED when the index exists or the search is
 * large enough to pay for building it. */
#define VSC_ALGO_AUTO 0
#define VSC_ALGO_SCAN 1
#define VSC_ALGO_SEED 2

/* Per-search device timings measured with HIP events on the context's stream (milliseconds). */
typedef struct {
    double scan_ms;          /* the dominant search kernel: scan_kernel or seed_sliced_kernel (last pass) */
    double prep_ms;          /* read upload (+ per-bucket read lists for VSC_ALGO_SEED) */
    double sort_ms;          /* bin sort of the hit records: histogram + partition level(s) */
    double finalize_ms;      /* ordering inside the bins + contig resolution + record assembly (+ deeper sort levels) */
    double score_ms;         /* last vsc_score_hits call */
    double total_ms;         /* first launch to last completion of the last vsc_search */
    double index_ms;         /* last seed-index build on this context */
    uint64_t sites;          /* PAM-valid, N-free windows of the shard (both strands) */
    uint64_t pairs;          /* (window, read) comparisons made */
    uint64_t hits;           /* hits reported */
    uint64_t genome_bytes;   /* genome bytes the search kernel streamed: planes (SCAN) or visited site records (SEED) */
    uint32_t passes;         /* search launches needed (1 per read pass unless a hit buffer overflowed) */
    uint32_t algorithm;      /* VSC_ALGO_SCAN or VSC_ALGO_SEED: what ran */
    uint64_t sort_bytes;     /* bytes the sort kernels moved (8 per packed record read or written, 16 per result record) */
    uint32_t sort_levels;    /* partition levels the bin sort ran (0: the regions fitted its last stage as they were) */
    uint32_t sort_bin_bits;  /* key bits of the first partition level */
    uint32_t read_passes;    /* passes over the read set (a pass takes at most 16 384 reads) */
    uint32_t sort_fallbacks; /* sort levels that started without a histogram pass (fixed bin slots) and had to run again with
                                one because a bin outgrew its slot (repeat-rich genomes; the genome remembers) */
    uint64_t list_entries;   /* VSC_ALGO_SEED: entries of the per-bucket read lists the last pass made (8 bytes each, padding included) */
    uint32_t seed_cut;       /* VSC_ALGO_SEED: k0 | k1 << 4 - substitutions within which read segments 0 and 1 were searched (the third:
                                what the site's PAM leaves of the limit - k0 - k1 - 2) */
    uint32_t reserved;
} vsc_timing;

/* ---- context ------------------------------------------------------------------------------- */
int vsc_abi_version(void);
/* Number of visible HIP devices (0 when there is none); never initialises a device. */
int vsc_device_count(void);
/* Binds device_id and creates the context's stream.  Replaces: process start-up of bidir_mapping
 * (read_mapping/bidir_mapping.cpp:190-258). */
int vsc_ctx_create(int device_id, vsc_ctx **out);
/* Genomes and results created on a context must be freed before it. */
int vsc_ctx_destroy(vsc_ctx *ctx);
/* Gives the context's pooled device memory back (search / sort / scoring scratch, record buffers of freed results -
 * tens of GB after a large search; they are kept because hipMalloc / hipFree of such sizes cost hundreds of
 * milliseconds).  Genomes and live results are untouched; the next call allocates what it needs again. */
int vsc_ctx_release_scratch(vsc_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. the framework's current stream) instead of the context's own. */
int vsc_ctx_set_stream(vsc_ctx *ctx, void *hip_stream);
/* Text of the last error on this context ("" if none).  Valid until the next call on ctx. */
const char *vsc_last_error(const vsc_ctx *ctx);
int vsc_ctx_timing(const vsc_ctx *ctx, vsc_timing *out);

/* ---- host-side packing helpers (no device needed) -------------------------------------------- */
/* Lays out n contigs in the global coordinate space (one N separator between contigs) and returns
 * the number of 32-bit words each plane needs.  Replaces the StringSet<Dna5String> concatenation of
 * read_mapping/bidir_index.cpp:36-40. */
uint64_t vsc_layout_contigs(const uint32_t *contig_len, uint32_t n_contigs, vsc_contig *out_table);
/* Initialises planes of n_words words to "all N". */
void vsc_planes_init(uint32_t *hi, uint32_t *lo, uint32_t *nmask, uint64_t n_words);
/* Writes n characters (ACGT any case; everything else = N, SeqAn Dna5 conversion) at global
 * position dst_pos. */
void vsc_pack_bases(const char *seq, uint64_t n, uint64_t dst_pos, uint32_t *hi, uint32_t *lo, uint32_t *nmask);
/* Inverse of vsc_pack_bases: n characters from global position src_pos. */
void vsc_unpack_bases(const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t src_pos, uint64_t n,
                      char *out);
/* Packs one 23-nt read: 2 bits per base, base i in bits 2i..2i+1, A=0 C=1 G=2 T=3; every other
 * character becomes A (SeqAn Dna conversion of the reads, read_mapping/bidir_mapping.cpp:194,256,264). */
uint64_t vsc_pack_guide(const char *seq23);

/* ---- genome ---------------------------------------------------------------------------------- */
/*
 * Uploads (a shard of) the packed genome.  The arrays hold words [first_word, first_word+n_words)
 * of the global planes; windows STARTING in the first own_words words are searched, the remaining
 * words are halo (a 22-base halo = 1 word is enough); positions past the arrays are N.
 * contigs / n_contigs always describe the WHOLE genome (positions are reported per contig).
 * The caller keeps ownership of the host arrays; the library owns the device copies.
 * Replaces: open(index, path) + contig-name pass of read_mapping/bidir_mapping.cpp:268-280 (the
 * FM index is replaced by the resident planes).
 */
int vsc_genome_load(vsc_ctx *ctx, const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t first_word,
                    uint64_t n_words, uint64_t own_words, const vsc_contig *contigs, uint32_t n_contigs,
                    vsc_genome **out);
int vsc_genome_free(vsc_genome *genome);
/* Builds (or keeps, if it matches) the seed index of a resident genome for the PAM set of `params`
 * (NULL = GG, GA only): the PAM-valid, N-free windows of both strands, filed once per 7-base
 * segment in bucket order (an 8-byte record + 4 bytes of bit-sliced planes per window and segment:
 * 36 bytes per window).  Plays the part of `bidir_index`
 * (read_mapping/bidir_index.cpp:45-47); vsc_search builds it on demand. */
int vsc_genome_build_index(vsc_ctx *ctx, vsc_genome *genome, const vsc_search_params *params);
/* The seed index as a file, for callers that keep `bidir_index`'s contract of an index built once and loaded by
 * every later run (read_mapping/bidir_index.cpp:45-47 writes, bidir_mapping.cpp:96-99 opens): _save writes the
 * resident index of `genome` (VSC_ERR_INVALID if it has none), _load replaces the genome's index with the file's.
 * The file names the library version, the PAM set, a fingerprint of the genome and the sizes of its arrays; _load
 * refuses a file that does not match or whose length is not what its header announces (VSC_ERR_INVALID,
 * vsc_last_error says why; the genome's index stays as it was - only a read error half-way leaves it without one).  36 bytes per window:
 * at 3 Gbp the file is 27 GB and rebuilding on the device (0.19 s) is faster than reading it - the tools only use the
 * file when asked to (bidir_index -S). */
int vsc_genome_index_save(vsc_ctx *ctx, const vsc_genome *genome, const char *path);
int vsc_genome_index_load(vsc_ctx *ctx, vsc_genome *genome, const char *path);
/* Bytes of HBM the resident genome (planes + seed index) occupies. */
uint64_t vsc_genome_device_bytes(const vsc_genome *genome);

/* ---- search ---------------------------------------------------------------------------------- */
/*
 * Searches every guide on both strands and returns all candidate sites, i.e. for every read the
 * union of the two searchAndVerifyEntireRead calls of read_mapping/bidir_mapping.cpp:285-295
 * (pigeonhole halves :157-162, find<0,k> :129-146, verify delegate :39-126), as the predicate
 * DESIGN.md states.  guides = n_guides values of vsc_pack_guide.  The result is sorted ascending
 * by (guide, strand '+' before '-', contig, pos) - the order of the reference's std::map (:154)
 * per read and strand; primary/secondary selection (:167-187) is host-side formatting
 * (vsc_sam_order).  The result is library-owned; release it with vsc_hits_free.
 */
int vsc_search(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
               const vsc_search_params *params, vsc_hits **out);
/*
 * The same search for read sets whose result does not fit the device at once (100 000 reads at 8
 * mismatches on 3 Gbp are 1.6e10 records): the reads are searched in batches of batch_reads (0 or more
 * than 16 384: 16 384) and every batch's result - same record order, guide = index into `guides` - is
 * handed to on_batch and freed when it returns.  The callback may use the result with any vsc_hits_* /
 * vsc_score_hits* call on the same context; a non-zero return value stops the stream and is returned.
 * Replaces the OpenMP loop over reads of read_mapping/bidir_mapping.cpp:285-295 for workloads where the
 * reference appends every read's records to its output buffer and moves on.
 * vsc_ctx_timing afterwards reports sums over the batches (score_ms: the callbacks' scoring calls).
 */
typedef int (*vsc_batch_fn)(void *user, vsc_hits *batch, uint32_t first_guide, uint32_t n_guides);
int vsc_search_stream(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
                      const vsc_search_params *params, uint32_t batch_reads, vsc_batch_fn on_batch, void *user);
/*
 * vsc_search_stream with the per-hit feature rows made ON THE WAY (BASELINE configuration 5: "streamed ... + per-hit
 * classification feature scoring"): every batch arrives with its 64-byte packed rows - rows_dev: device memory,
 * vsc_hits_count(batch) * 64 bytes, row i belongs to record i, valid inside the callback; the same rows as
 * vsc_score_hits_packed writes.  The search keeps each site's bases beside its hit record (4 bytes per hit through the
 * sort) and the kernel that assembles the vsc_hit records writes the row in the same pass - instead of a second kernel
 * that re-reads the records and gathers 23 bases per hit from the planes (a 64-byte sector for 46 bits).
 * Replaces, per batch: read_mapping/bidir_mapping.cpp:285-295 + variant_processing/merge_output_bam.h:696-708
 * (featureMatrixRecord of every record, feature_matrix.h:25-126).
 */
typedef int (*vsc_rows_batch_fn)(void *user, vsc_hits *batch, uint32_t first_guide, uint32_t n_guides, const void *rows_dev);
int vsc_search_stream_rows(vsc_ctx *ctx, const vsc_genome *genome, const uint64_t *guides, uint32_t n_guides,
                           const vsc_search_params *params, uint32_t batch_reads, vsc_rows_batch_fn on_batch, void *user);
uint64_t vsc_hits_count(const vsc_hits *hits);
/* Device pointer to vsc_hits_count() records of type vsc_hit (valid until vsc_hits_free). */
const void *vsc_hits_data_dev(const vsc_hits *hits);
/* Host copy of the records (made on first use; valid until vsc_hits_free). */
int vsc_hits_data(vsc_hits *hits, const vsc_hit **out);
/* Copies the records into caller memory (host, or device when dst_is_device != 0 - e.g. a tensor the
 * caller hands to RCCL). */
int vsc_hits_copy(vsc_hits *hits, void *dst, int dst_is_device);
/*
 * Multi-GPU: merges the results of genome shards.  records = the vsc_hit records of shard 0,
 * shard 1, ... concatenated in shard order (shard_counts[s] records each, as vsc_search returned them;
 * contig and pos are already global, so nothing is rewritten), in device memory when
 * records_on_device != 0 (the buffer RCCL gathered into), else in host memory (uploaded first).
 * Shards partition the positions in ascending order, so the global result is, for every
 * (guide, strand), shard 0's segment followed by shard 1's, ...: segments are located and copied,
 * nothing is sorted.  Replaces the concatenation of per-thread output buffers,
 * read_mapping/bidir_mapping.cpp:307-308.
 */
int vsc_hits_merge(vsc_ctx *ctx, const void *records, int records_on_device, const uint64_t *shard_counts,
                   uint32_t n_shards, uint32_t n_guides, vsc_hits **out);
/*
 * The same merge over the 8-byte EXCHANGE RECORDS - what the multi-GPU drivers send over xGMI instead of the
 * 16-byte vsc_hit: bits 0..22 mismatch mask | bits 23..54 global position (contig offset + pos).  Guide and strand
 * are not sent: a shard's records are sorted by key = guide << 1 | strand, so 4 bytes per key (key_counts) say which
 * records belong to which key; NM is the popcount of the mask; the contig follows from the global position.
 *   vsc_hits_pack_exchange  writes vsc_hits_count(hits) records (device memory when records_on_device != 0, e.g. a
 *                           tensor handed to RCCL) and the 2 * n_guides per-key record counts (host memory).
 *   vsc_hits_merge_packed   records = the exchange records of shard 0, shard 1, ... concatenated, each shard's for the
 *                           keys [first_key, first_key + n_keys) only (a receiver may collect just its read range);
 *                           key_counts[s * n_keys + k] = records of shard s with key first_key + k.  The result holds
 *                           ordinary vsc_hit records (guide = key >> 1), sorted as vsc_search sorts them; `genome`
 *                           (any shard of the genome on this context) supplies the contig table.
 */
#define VSC_XREC_BYTES 8
int vsc_hits_pack_exchange(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, uint32_t n_guides, void *records,
                           int records_on_device, uint32_t *key_counts);
int vsc_hits_merge_packed(vsc_ctx *ctx, const vsc_genome *genome, const void *records, int records_on_device,
                          const uint32_t *key_counts, uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out);
/* vsc_hits_merge_packed for records that carry 2 bytes of per-hit results each (the votes of the shard's classifier,
 * vsc_score_classify_hits: computed where the hit was found, before the exchange): `votes` = one uint16 per exchange record, same
 * order and the same memory space as `records`; votes_out (count of the merged records x 2 bytes; device memory when
 * votes_out_on_device != 0) receives them in the order of the merged records. */
int vsc_hits_merge_packed_votes(vsc_ctx *ctx, const vsc_genome *genome, const void *records, const void *votes, int on_device,
                                const uint32_t *key_counts, uint32_t n_shards, uint32_t first_key, uint32_t n_keys, vsc_hits **out,
                                void *votes_out, int votes_out_on_device);
int vsc_hits_free(vsc_hits *hits);

/*
 * Per-hit scores for rows [first, first+count) of a result (rows R5/R6):
 *   mit       : count doubles  = calcMitScore (variant_processing/mit_score.h:12-68) on the hit's
 *               mismatch positions (forward-genome orientation, as merge_output_bam.h:549 passes them)
 *   mit_flags : count bytes, 1 where the reference would index its weight table out of bounds
 *   features  : count * 442 bytes = featureMatrixRecord (variant_processing/feature_matrix.h:25-126)
 *               of (guide, off-target in guide orientation), as merge_output_bam.h:696 calls it
 * Any of the three output pointers may be NULL.  guides must be the array passed to vsc_search.
 * A row range whose scores do not fit the free device memory at once is scored in several passes over
 * the same scratch buffers (VSC_ERR_NOMEM only if not even 65 536 rows fit).
 */
int vsc_score_hits(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, const uint64_t *guides,
                   uint32_t n_guides, uint64_t first, uint64_t count, double *mit, uint8_t *mit_flags,
                   uint8_t *features);
#define VSC_N_FEATURES 442
/*
 * Feature rows in packed form, 64 bytes (16 little-endian 32-bit words) per hit instead of 442:
 *   w0      bits 0..20 mismatchPos1..21 | 21..25 totalMismatches | 26..30 adjacentMismatches
 *   w1      bits 0..11 AtoC..TtoG | 12..16 transitionNumber | 17..21 transversionNumber | 22..25 seedMismatches
 *   w2..w4  A1..T20, PAMA..PAMT one-hots (bit 4 i + base)
 *   w5..w14 AA1..TT19 one-hots (bit 16 i + pair); the 16 dinucleotide counts are their column sums
 *   w15     0
 * The rows are written to packed_dev (device memory, count * 64 bytes) or, if that is NULL, to library
 * scratch - as many rows per pass as fit the free device memory (a 10 000-read result at 8 mismatches is
 * 1.6e9 rows = 104 GB) - and, if packed_host is not NULL, copied to the host pass by pass; with neither
 * destination the rows are computed and dropped (timing runs).  mit_host (optional) receives the MIT scores.
 * vsc_unpack_features expands rows to the dense 442-byte form of vsc_score_hits.
 */
int vsc_score_hits_packed(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, const uint64_t *guides,
                          uint32_t n_guides, uint64_t first, uint64_t count, void *packed_dev, uint32_t *packed_host,
                          double *mit_host);
#define VSC_PACKED_FEATURE_BYTES 64
void vsc_unpack_features(const uint32_t *packed, uint64_t n, uint8_t *features);
/*
 * The same scores for n explicit (on-target, off-target) pairs: both 23-mers as vsc_pack_guide codes
 * in read orientation, masks[i] = the mismatch positions calcMitScore is given for row i (bit p =
 * position p).  This is the form the mergers need: they score what they re-derived from the SAM
 * records (variant_processing/merge_output_bam.h:156,187,389,437,549,696), where the positions
 * come from the MD tag (filter_output_bam.h:330-349) and the sequences from the FASTA (:399,484).
 */
int vsc_score_pairs(vsc_ctx *ctx, const uint64_t *on_targets, const uint64_t *off_targets, const uint32_t *masks,
                    uint64_t n, double *mit, uint8_t *mit_flags, uint8_t *features);

/* ---- several devices of one node ----------------------------------------------------------------------- */
/*
 * The genome-sharded search of one host process over n devices (BASELINE north star: "partitioned across
 * the 8 GPUs of one node by genome shard with a single RCCL gather of candidate hits"): one vsc_ctx per
 * entry of device_ids, the planes cut into n tile-aligned position ranges (+ a one-word halo; a window
 * belongs to the shard that holds its first base), every device searching ALL reads on its shard from its own
 * host thread, then one exchange - the records (8-byte exchange records, vsc_hits_pack_exchange) by one grouped
 * ncclSend / ncclRecv per shard to the first device, over xGMI (the counts are known inside the process) - and
 * vsc_hits_merge_packed there.  The result is an
 * ordinary vsc_hits of a context on the first device (vsc_multi_result_ctx(m)): same records, same order as one device gives.
 * This replaces, inside the bidir_mapping process, the OpenMP loop over reads and the concatenation of the
 * per-thread output buffers (read_mapping/bidir_mapping.cpp:285-295,307-308): the parallel axis is the genome.
 * RCCL is bound at run time (dlopen) and used when n > 1 distinct devices are given; device ids may repeat
 * (several contexts on one GPU: tests and rehearsals on a one-GPU box) - the exchange then is device copies.
 * vsc_multi_last_error after vsc_multi_create says why copies are in use when RCCL could not be set up.
 */
typedef struct vsc_multi vsc_multi;
typedef struct vsc_multi_genome vsc_multi_genome;
typedef struct {
    double search_wall_ms;   /* host wall time until the slowest shard had searched (+ scored) and packed its last batch */
    double search_ms_max;    /* largest sum of vsc_timing.total_ms among the shards */
    double exchange_ms;      /* host wall time the exchange took beyond the searches: from the moment the last shard of a batch was
                                ready until that batch's records had arrived on the first device (summed over the batches) */
    double merge_ms;         /* vsc_hits_merge_packed on the first device (host wall, summed over the batches) */
    double total_ms;
    uint64_t hits;
    uint64_t exchanged_bytes; /* record (+ vote) bytes that crossed between devices */
    uint32_t n_devices;
    uint32_t used_rccl;      /* 1: RCCL carried the exchange, 0: device copies */
    double score_ms_max;     /* largest sum of the shards' scoring kernels (vsc_timing.score_ms) */
    double callback_ms;      /* host wall time inside on_batch (summed) */
    uint32_t batches;        /* batches the reads were searched in (vsc_multi_search: 1) */
    uint32_t reserved;
} vsc_multi_timing;
int vsc_multi_create(const int *device_ids, int n, vsc_multi **out);
int vsc_multi_destroy(vsc_multi *m);
/* vsc_ctx_release_scratch on every context of the set + the pooled exchange buffers (they stay allocated between searches). */
int vsc_multi_release_scratch(vsc_multi *m);
int vsc_multi_size(const vsc_multi *m);
vsc_ctx *vsc_multi_ctx(vsc_multi *m, int i);  /* the i-th shard's context */
/* The context on the first device that owns the merged results (its own stream: a batch is merged there while the shards -
 * the first device's among them - search the next one). */
vsc_ctx *vsc_multi_result_ctx(vsc_multi *m);
const char *vsc_multi_last_error(const vsc_multi *m);
int vsc_multi_uses_rccl(const vsc_multi *m);
int vsc_multi_get_timing(const vsc_multi *m, vsc_multi_timing *out);
/* hi / lo / nmask: the WHOLE genome's planes (n_words words each); every device receives its shard. */
int vsc_multi_genome_load(vsc_multi *m, const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t n_words,
                          const vsc_contig *contigs, uint32_t n_contigs, vsc_multi_genome **out);
int vsc_multi_genome_free(vsc_multi_genome *g);
int vsc_multi_genome_build_index(vsc_multi *m, vsc_multi_genome *g, const vsc_search_params *params);
/* as vsc_search; *out belongs to vsc_multi_result_ctx(m) and is released with vsc_hits_free.  A shard's records leave for
 * the first device as soon as THAT shard is done (they do not wait for the slowest one). */
int vsc_multi_search(vsc_multi *m, const vsc_multi_genome *g, const uint64_t *guides, uint32_t n_guides,
                     const vsc_search_params *params, vsc_hits **out);
/* (vsc_multi_search_stream, which scores on the owning shard, is declared behind the classifier below.) */

/* ---- variant windows (row R8) ------------------------------------------------------------------- */
/*
 * The alt-allele windows of one sample column of a VCF ("SNP genome"), built straight into packed planes:
 * what `vcf_loader FILE.vcf SNP.fa GENOME.fa SAMPLE SEQLENGTH THREADS` followed by `bidir_index -G SNP.fa`
 * produce (variant_processing/vcf_loader.cpp:40-68, process_vcf.h:54-269, overlap_sequences.h:35-240,
 * write_fasta.h:30-470; VARSCOT:296-307) - same windows, same order, same ids, planes byte-identical -
 * without the FASTA text in between.  Reference segments are copied bit-wise from the reference planes
 * hi / lo / nmask (whole genome, host memory; contigs / contig_names describe it, a chromosome is found by
 * the first word of its name as with an FAI index) instead of per-segment FAI reads (write_fasta.h:245-271);
 * parsing, sweep and window assembly run on `threads` host threads (0 = all).  Host-only, no device needed.
 * The result is a genome like any other: vsc_genome_load(planes, contigs) + vsc_search.  One call per sample
 * column replaces one run of the reference pipeline per sample (parallel.py:49-63); the reference genome and
 * its search are shared between the samples.
 * err (optional, err_len bytes) receives the text of a failure.
 */
typedef struct vsc_windows vsc_windows;
int vsc_windows_build(const char *vcf_path, uint32_t sample, uint32_t seq_len, uint32_t threads, const uint32_t *hi,
                      const uint32_t *lo, const uint32_t *nmask, const vsc_contig *contigs, const char *const *contig_names,
                      uint32_t n_contigs, vsc_windows **out, char *err, size_t err_len);
uint32_t vsc_windows_count(const vsc_windows *w);            /* windows = contigs of the SNP genome */
uint64_t vsc_windows_words(const vsc_windows *w);            /* 32-bit words per plane */
const uint32_t *vsc_windows_plane(const vsc_windows *w, int which); /* 0 hi, 1 lo, 2 nmask */
const vsc_contig *vsc_windows_contigs(const vsc_windows *w);
/* id of window i (chr_start_REF / chr_start_ALT_pos_ref_alt..., write_fasta.h:30-65); not NUL-terminated */
const char *vsc_windows_name(const vsc_windows *w, uint32_t i, uint32_t *len);
/* count + 1 offsets into the id pool that starts at vsc_windows_name(w, 0, NULL); every id is followed by '\n' */
const uint64_t *vsc_windows_name_offsets(const vsc_windows *w);
void vsc_windows_free(vsc_windows *w);

/* ---- classifier ---------------------------------------------------------------------------------- */
/*
 * A trained random forest as randomForest stores it ($forest of the object in
 * classification/rfClassifier.RData), numeric splits only, two classes ("0", "1").  All arrays have
 * n_trees * n_nodes entries, tree-major.  feature[] = column of the dense feature row the node
 * tests (0..441, or 442 = the on-target activity).
 */
typedef struct {
    uint32_t n_trees, n_nodes;
    const int8_t *node_status;  /* 1 = split node, -1 = terminal */
    const uint16_t *feature;
    const uint16_t *left, *right; /* 1-based daughter nodes */
    const double *split;        /* x <= split goes left */
    const uint8_t *node_class;  /* terminal nodes: 1 = class "0", 2 = class "1" */
} vsc_rf_model;
/*
 * predict(rfClassifier, featureMatrix[, type = "prob"]) of classification/classificationPipeline.R:27-34
 * for n feature rows (dense 442-byte rows as vsc_score_hits / vsc_score_pairs produce them, plus the
 * on-target activity of each row): prob = share of trees voting class "1", cls = 1 if that share is
 * above one half, tie = 1 where the vote is exactly split (R breaks such ties at random).
 */
int vsc_rf_predict(vsc_ctx *ctx, const vsc_rf_model *model, const uint8_t *features, const double *activity, uint64_t n,
                   double *prob, uint8_t *cls, uint8_t *tie);
/* The same for rows in the packed 64-byte form of vsc_score_hits_packed (host memory, or device memory when
 * rows_on_device != 0 - e.g. the buffer vsc_score_hits_packed has just filled): the columns the forest tests
 * are decoded on the device, the 442-byte rows never exist. */
int vsc_rf_predict_packed(vsc_ctx *ctx, const vsc_rf_model *model, const void *packed_rows, int rows_on_device,
                          const double *activity, uint64_t n, double *prob, uint8_t *cls, uint8_t *tie);

/*
 * Score -> classify in one kernel, for results whose feature rows should never exist in memory (100 000 reads at 8
 * mismatches are 1.6e10 hits = 1 TB of packed rows): for rows [first, first + count) of a search result the kernel
 * computes the hit's feature row in registers (as vsc_score_hits_packed would write it), walks the forest and writes
 * the number of trees voting class "1" - 2 bytes per hit (prob = votes / n_trees, class = 2 * votes > n_trees,
 * tie = 2 * votes == n_trees), and optionally the MIT score.  guide_activity[g] = the on-target activity of read g
 * (the column classification/classificationPipeline.R:21-25 reads from the feature file, constant per target;
 * variant_processing/merge_output_bam.h:401,708 appends it to every row).  Identical, hit for hit, to
 * vsc_score_hits_packed followed by vsc_rf_predict_packed.  votes_dev (device, count * 2 bytes) and votes_host are
 * optional destinations; with neither the votes stay in library scratch (timing runs).
 * Replaces, for a streamed search: variant_processing/merge_output_bam.h:696-708 (feature rows as text) +
 * classification/classificationPipeline.R:21-49 (read them back, predict).
 */
int vsc_score_classify_hits(vsc_ctx *ctx, const vsc_genome *genome, const vsc_hits *hits, const uint64_t *guides, uint32_t n_guides,
                            const double *guide_activity, const vsc_rf_model *model, uint64_t first, uint64_t count, void *votes_dev,
                            uint16_t *votes_host, double *mit_host);

/* ---- several devices, streamed + scored on the owning shard ------------------------------------------------ */
/*
 * The streamed search of vsc_search_stream over the device set (BASELINE configuration 5: "100 000 guides streamed ... +
 * per-hit classification feature scoring, 8 x MI355X"): the reads go through all shards batch by batch; what `score` asks
 * for is computed per hit ON THE SHARD THAT FOUND IT, before the exchange; a batch's records (+ votes) travel to the first
 * device and are merged there WHILE the shards search the next batch (two exchange buffers per shard, a merge context of its
 * own on the first device); on_batch receives every merged batch - same records, same order, same read indices as one
 * device's vsc_search_stream gives - and the batch is freed when it returns.
 *   VSC_MULTI_SCORE_NONE   nothing
 *   VSC_MULTI_SCORE_ROWS   the 64-byte packed feature rows of vsc_score_hits_packed, computed and dropped on the shard (what
 *                          a consumer on the shard would read; a timing run like the one-device c5 bench)
 *   VSC_MULTI_SCORE_VOTES  vsc_score_classify_hits: the forest's votes, 2 bytes per hit, travel with the 8-byte record and
 *                          arrive merged: votes_dev[i] belongs to record i of the batch (device memory of the first device)
 * Replaces read_mapping/bidir_mapping.cpp:285-295,307-308 (the loop over reads and the concatenation of the per-thread
 * buffers) + variant_processing/merge_output_bam.h:696-708 + classification/classificationPipeline.R:21-49 for a read set
 * whose result fits no device.
 */
#define VSC_MULTI_SCORE_NONE 0
#define VSC_MULTI_SCORE_ROWS 1
#define VSC_MULTI_SCORE_VOTES 2
typedef struct {
    uint32_t mode;                 /* VSC_MULTI_SCORE_* */
    uint32_t reserved;
    const double *guide_activity;  /* VOTES: on-target activity per read (n_guides values) */
    const vsc_rf_model *model;     /* VOTES: the forest */
} vsc_multi_score;
typedef int (*vsc_multi_batch_fn)(void *user, vsc_hits *batch, uint32_t first_guide, uint32_t n_guides, const uint16_t *votes_dev);
int vsc_multi_search_stream(vsc_multi *m, const vsc_multi_genome *g, const uint64_t *guides, uint32_t n_guides,
                            const vsc_search_params *params, uint32_t batch_reads, const vsc_multi_score *score,
                            vsc_multi_batch_fn on_batch, void *user);

/* ---- host-side formatting helpers (no device needed) ------------------------------------------ */
/*
 * Order in which read_mapping/bidir_mapping.cpp:167-187 writes the records of one search result
 * and which of them it flags BAM_FLAG_SECONDARY.  hits must be sorted as vsc_search returns them.
 * order[i] = index into hits of the i-th record written; secondary[i] = 1 if that record carries
 * flag 256.
 */
void vsc_sam_order(const vsc_hit *hits, uint64_t n, uint64_t *order, uint8_t *secondary);
#ifdef __cplusplus
}
#endif
#endif
