/*
 * vsc_oracle.h - CPU restatement of VARSCOT's off-target search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under varscot_amd/ may include, link, import or execute
 * anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use
 * it, and only as the checker / the timed CPU baseline - never as the thing shipped.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - search (rows R1-R4): the search primitive of the reference lives in SeqAn 2.4.0rc2
 *     (VARSCOT_pipeline/Dockerfile:41), which is not under /root/reference and cannot be fetched; the
 *     reference's own SAM/TSV outputs are absent git-LFS blobs.  The restatement is derived from the
 *     source of read_mapping/bidir_mapping.cpp (cited per function) and PINNED where the reference
 *     still holds anything: acceptance, strand and NM on the 2 779 (guide, site, NM) triples of
 *     VARSCOT's own SAM output kept in workflow/data-objects/datasetsSampling.RData, the write order
 *     on the 348 SAM row numbers of indexGuideSeq.RData (tests/test_oracle.py).  UNPINNED (nothing in
 *     the reference can pin them): completeness of the hit set, the right-edge rule, N handling, the
 *     MD text.
 *   - feature matrix (row R6): pinned by tests/golden/features_golden.npz (6960 x 442 values from
 *     workflow/data-objects/featureMatrix.RData).
 *   - MIT score (row R5): pinned by the known answers SURVEY.md section 8 R5 lists and, where the
 *     two formulas coincide, by 2425 rows of workflow/pipeline-comparison/crispor-siteseq-offtargets.txt.
 *
 * All file:line citations are relative to /root/reference/VARSCOT_pipeline/.
 */
#ifndef VSC_ORACLE_H
#define VSC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_READ_LEN 23

/* Same 16-byte layout as vsc_hit in include/varscot_hip.h. */
typedef struct {
    uint32_t guide;  /* index of the guide in input order */
    uint32_t contig; /* index of the contig in FASTA order */
    uint32_t pos;    /* 0-based start of the 23-base window on the forward strand */
    uint32_t info;   /* bit31 strand (1 = '-'), bit30 secondary, bits 23..27 NM, bits 0..22 mismatch
                        mask in forward-genome window coordinates (bit i = window position i) */
} orc_hit;

#define ORC_INFO_STRAND(i) (((i) >> 31) & 1u)
#define ORC_INFO_SECONDARY(i) (((i) >> 30) & 1u)
#define ORC_INFO_NM(i) (((i) >> 23) & 31u)
#define ORC_INFO_MASK(i) ((i) & 0x7FFFFFu)

enum {
    ORC_MODE_PREDICATE = 0, /* brute force over every window, SURVEY.md section 8.1 */
    ORC_MODE_REFERENCE_FLOW = 1 /* follows the control flow of read_mapping/bidir_mapping.cpp */
};

/*
 * Search.  contigs[c] points at contig_len[c] characters (any case; everything except ACGT is N,
 * as SeqAn's Dna5 conversion does).  guides = n_guides * 23 characters, no separators (everything
 * except ACGT becomes A, as SeqAn's Dna conversion does - bidir_mapping.cpp:256,264).
 * extra_pam: NULL or 2 characters (the -P option, bidir_mapping.cpp:216,242-247).
 * compat_u16: reproduce the uint16_t contig truncation of the dedup key (bidir_mapping.cpp:13).
 *
 * Output order: ORC_MODE_PREDICATE -> ascending (guide, strand, contig, pos), no secondary flags;
 * ORC_MODE_REFERENCE_FLOW -> the order in which bidir_mapping.cpp:167-187 writes the records,
 * with the secondary flag set as it sets BAM_FLAG_SECONDARY.
 * Returns the number of hits found (which can exceed cap; only the first cap are stored), or -1.
 */
long orc_search(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs,
                const char *guides, uint32_t n_guides, uint32_t max_mm, const char *extra_pam,
                int mode, int compat_u16, orc_hit *out, long cap);

/* read_mapping/bidir_mapping.cpp:88-123 + SURVEY.md 8.5: the header-less SAM text of a search in
 * ORC_MODE_REFERENCE_FLOW order.  md_style 0 = SAM-spec MD (zeros between adjacent mismatches and
 * at both ends), 1 = no zeros.  Returns a malloc'd NUL-terminated string. */
char *orc_search_sam(const char *const *contigs, const uint32_t *contig_len,
                     const char *const *contig_names, uint32_t n_contigs, const char *guides,
                     const char *const *guide_names, uint32_t n_guides, uint32_t max_mm,
                     const char *extra_pam, int md_style);
void orc_free(void *p);

/* MD:Z value for a 23-base window vs read (reference bases at mismatching positions). */
void orc_md_string(const char *window, const char *read, int md_style, char *out /* >= 64 */);

/* variant_processing/filter_output_bam.h:330-349: positions the reference recovers from an MD
 * string.  Returns the count; an empty result is reported as the single value -1. */
int orc_md_positions(const char *md, int *out /* >= 24 */);

/* variant_processing/mit_score.h:12-68.  positions = mismatch positions (ascending) or the single
 * value -1 for a perfect match.  *ub is set when the reference would index matrixM out of bounds
 * (SURVEY.md 8.2 Q2); such an index contributes the factor (1 - 0). */
double orc_mit_score(const int *positions, int n, int *ub);

/* variant_processing/feature_matrix.h:25-126: the 442 sequence features. */
void orc_feature_row(const char *on_target, const char *off_target, uint32_t *features /* 442 */);

/* Bit-parallel OpenMP port of the same predicate for sizes the char-based functions above would
 * take minutes on (vsc_fastport.c).  Same contract as orc_search in ORC_MODE_PREDICATE. */
long orc_search_fast(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs,
                     const char *guides, uint32_t n_guides, uint32_t max_mm, const char *extra_pam,
                     int threads, orc_hit *out, long cap);
/* Count-only variant used by bench.py's cpu_baseline leg: sites = PAM-valid, N-free windows
 * compared (both strands). */
long orc_count_fast(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs,
                    const char *guides, uint32_t n_guides, uint32_t max_mm, const char *extra_pam,
                    int threads, long *sites);
int orc_max_threads(void);

/* vsc_planes.c: n characters from global position pos of a genome in the packed-plane layout of
 * include/varscot_hip.h (restated from the header, independent of the product's unpacking). */
void orc_planes_to_text(const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t pos, uint64_t n, char *out);

/* The same search in the reference's algorithmic shape (vsc_pigeon.c): pigeonhole halves with floor(m/2)
 * substitutions each (read_mapping/bidir_mapping.cpp:129-146,157-162) through a k-mer table of the text,
 * every occurrence verified by the delegate (:39-126), reads under OpenMP (:285-295).  Build once per text. */
typedef struct orc_pigeon orc_pigeon;
orc_pigeon *orc_pigeon_build(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs);
void orc_pigeon_free(orc_pigeon *ix);
/* Returns the number of hits (stores the first cap, sorted by (guide, strand, contig, pos), when out != NULL);
 * *candidates = occurrences the delegate was called on. */
long orc_pigeon_search(const orc_pigeon *ix, const char *reads, uint32_t n_reads, uint32_t max_mm, const char *extra_pam,
                       int threads, orc_hit *out, long cap, long *candidates);

#ifdef __cplusplus
}
#endif
#endif
