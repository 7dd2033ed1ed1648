/*
 * vsc_pigeon.c - CPU port of the reference's search in the reference's own algorithmic shape
 * (VARSCOT_pipeline/read_mapping/bidir_mapping.cpp): per read and strand, the pigeonhole split into a first
 * half read[0,11) and a second half read[11,23) (:157-162), each half searched with at most k = floor(m/2)
 * substitutions (:129-146), every occurrence extended to the full 23-mer and verified by the delegate
 * (:39-126: room to the right / left, already-recorded check, PAM, N, Hamming distance <= m); reads are
 * independent and run under OpenMP like the reference's loop (:285-295).
 *
 * TEST INFRASTRUCTURE ONLY - see vsc_oracle.h: used by tests/ (against vsc_oracle.c) and by bench.py's
 * cpu_baseline leg ("kind": "port").  Search parity is UNPINNED like the rest of the search oracle.
 *
 * What stands in for SeqAn's bidirectional FM index (find<0,k>, absent from /root/reference): a k-mer table
 * of the text - all 11-mers and all 12-mers, positions grouped by k-mer - and an enumeration of every
 * 11- / 12-mer within k substitutions of the read half.  An FM-index backtracking search visits exactly
 * those strings that occur in the text; on a genome of more than 4^12 bases that is nearly all of them, so
 * the candidate set (what the delegate is called on) is the same and the work has the same shape:
 * enumerate <= k-error half-matches, look up their occurrences, verify each.
 */
#include "vsc_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RL ORC_READ_LEN
#define H1 (RL / 2)      /* first half: 11 */
#define H2 (RL - RL / 2) /* second half: 12 */

typedef struct orc_pigeon {
    uint8_t *code;        /* one byte per text position: 0..3 = ACGT, 4 = N / separator */
    uint64_t n;           /* text length (contigs joined by one separator) */
    uint32_t n_contigs;
    uint64_t *contig_off; /* start of every contig in the text */
    uint32_t *contig_len;
    uint32_t *start[2];   /* [4^L + 1] first entry of every L-mer in pos[], L = 11, 12 */
    uint32_t *pos[2];     /* text positions grouped by L-mer */
} orc_pigeon;

static int base_code(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

static int build_table(orc_pigeon *ix, int which, int L)
{
    const uint64_t buckets = 1ULL << (2 * L);
    const uint64_t mask = buckets - 1;
    uint32_t *start = (uint32_t *)calloc(buckets + 1, sizeof(uint32_t));
    if (!start) return -1;
    /* pass 1: count (rolling code; `valid` = bases since the last N) */
    for (int pass = 0; pass < 2; ++pass) {
        uint64_t code = 0;
        int valid = 0;
        for (uint64_t i = 0; i < ix->n; ++i) {
            const int c = ix->code[i];
            if (c > 3) {
                valid = 0;
                continue;
            }
            code = ((code << 2) | (uint64_t)c) & mask;
            if (++valid >= L) {
                if (pass == 0)
                    start[code + 1]++;
                else
                    ix->pos[which][start[code]++] = (uint32_t)(i + 1 - L);
            }
        }
        if (pass == 0) {
            for (uint64_t b = 0; b < buckets; ++b) start[b + 1] += start[b];
            ix->pos[which] = (uint32_t *)malloc(((size_t)start[buckets] + 1) * sizeof(uint32_t));
            if (!ix->pos[which]) {
                free(start);
                return -1;
            }
        }
    }
    /* pass 2 advanced start[b] to the end of bucket b = the start of bucket b + 1: shift back */
    memmove(start + 1, start, buckets * sizeof(uint32_t));
    start[0] = 0;
    ix->start[which] = start;
    return 0;
}

/* text = the contigs, in order; the first base of the k-mer code is its most significant digit */
orc_pigeon *orc_pigeon_build(const char *const *contigs, const uint32_t *lens, uint32_t n_contigs)
{
    orc_pigeon *ix = (orc_pigeon *)calloc(1, sizeof(orc_pigeon));
    if (!ix) return NULL;
    uint64_t n = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) n += (uint64_t)lens[c] + 1;
    if (n >= 0xFFFFFFFFULL) {
        free(ix);
        return NULL;
    }
    ix->n = n;
    ix->n_contigs = n_contigs;
    ix->code = (uint8_t *)malloc(n + 1);
    ix->contig_off = (uint64_t *)malloc(((size_t)n_contigs + 1) * sizeof(uint64_t));
    ix->contig_len = (uint32_t *)malloc(((size_t)n_contigs + 1) * sizeof(uint32_t));
    if (!ix->code || !ix->contig_off || !ix->contig_len) {
        orc_pigeon_free(ix);
        return NULL;
    }
    uint64_t at = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        ix->contig_off[c] = at;
        ix->contig_len[c] = lens[c];
        for (uint32_t i = 0; i < lens[c]; ++i) ix->code[at + i] = (uint8_t)base_code(contigs[c][i]);
        at += lens[c];
        ix->code[at++] = 4;
    }
    if (build_table(ix, 0, H1) || build_table(ix, 1, H2)) {
        orc_pigeon_free(ix);
        return NULL;
    }
    return ix;
}

void orc_pigeon_free(orc_pigeon *ix)
{
    if (!ix) return;
    free(ix->code);
    free(ix->contig_off);
    free(ix->contig_len);
    for (int w = 0; w < 2; ++w) {
        free(ix->start[w]);
        free(ix->pos[w]);
    }
    free(ix);
}

typedef struct {
    const orc_pigeon *ix;
    const uint8_t *read;   /* 23 codes of the searched strand (read or its reverse complement) */
    uint32_t guide;
    int strand;            /* 1 = the reverse complement is searched: hits are '-' */
    int max_mm, k;
    int first_half;
    int pam[3][2];         /* allowed PAMs as codes, in the orientation of the searched strand */
    int n_pam;
    orc_hit *out;          /* may be NULL (count only) */
    long cap, n_hits;
    long candidates;       /* occurrences the delegate was called on */
} search_state;

static uint32_t contig_of(const orc_pigeon *ix, uint64_t p)
{
    uint32_t lo = 0, hi = ix->n_contigs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (ix->contig_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

/* the delegate of bidir_mapping.cpp:39-126 for one occurrence of the searched half at text position `occ` */
static void verify(search_state *s, uint64_t occ)
{
    const orc_pigeon *ix = s->ix;
    s->candidates++;
    const uint32_t c = contig_of(ix, occ);
    const uint64_t cstart = ix->contig_off[c], clen = ix->contig_len[c];
    uint64_t p = occ - cstart; /* position in the contig */
    if (s->first_half) {
        if (clen <= p + RL) return; /* :51-52: no room to the right (a window ending AT the contig end is rejected too) */
    } else {
        if (p < (uint64_t)H1) return; /* :57-58: no room to the left */
        p -= H1;                      /* :61 */
        /* :64-65 records.find(): the first-half search has already recorded this window iff it accepted it - its
         * half within k substitutions and room to the right; everything else about the window is the same test */
        int d1 = 0;
        const uint8_t *w1 = ix->code + cstart + p;
        for (int i = 0; i < H1; ++i) d1 += w1[i] != s->read[i];
        if (d1 <= s->k && clen > p + RL) return;
    }
    const uint8_t *w = ix->code + cstart + p;
    /* PAM (:71-76): '+' strand window[21..23), '-' strand window[0..2) */
    int ok = 0;
    for (int q = 0; q < s->n_pam && !ok; ++q)
        ok = s->strand ? (w[0] == s->pam[q][0] && w[1] == s->pam[q][1]) : (w[RL - 2] == s->pam[q][0] && w[RL - 1] == s->pam[q][1]);
    if (!ok) return;
    /* N (:81-82) and mismatches over all 23 positions (:79-86).  On '-' the genome window is compared with the
     * reverse-complemented read, i.e. position i of the window with s->read[i] */
    unsigned mm = 0, mask = 0;
    for (int i = 0; i < RL; ++i) {
        if (w[i] > 3) return;
        if (w[i] != s->read[i]) {
            if (++mm > (unsigned)s->max_mm) return;
            mask |= 1u << i;
        }
    }
    if (s->out && s->n_hits < s->cap) {
        orc_hit h;
        h.guide = s->guide;
        h.contig = c;
        h.pos = (uint32_t)p;
        h.info = ((uint32_t)s->strand << 31) | (mm << 23) | mask; /* mask in forward-genome window coordinates */
        s->out[s->n_hits] = h;
    }
    s->n_hits++;
}

/* every L-mer within `left` substitutions of half[i..L), prefix code `code`: the strings find<0,k> walks */
static void enumerate(search_state *s, const uint8_t *half, int L, int i, int left, uint64_t code, int which)
{
    if (i == L) {
        const uint32_t *st = s->ix->start[which];
        for (uint32_t e = st[code]; e < st[code + 1]; ++e) verify(s, s->ix->pos[which][e]);
        return;
    }
    enumerate(s, half, L, i + 1, left, (code << 2) | half[i], which);
    if (left > 0)
        for (int b = 0; b < 4; ++b)
            if (b != half[i]) enumerate(s, half, L, i + 1, left - 1, (code << 2) | (uint64_t)b, which);
}

static int cmp_hit(const void *a, const void *b)
{
    const orc_hit *x = (const orc_hit *)a, *y = (const orc_hit *)b;
    if (x->guide != y->guide) return x->guide < y->guide ? -1 : 1;
    const unsigned sx = x->info >> 31, sy = y->info >> 31;
    if (sx != sy) return sx < sy ? -1 : 1;
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

/*
 * Searches n_reads reads (23 characters each, concatenated; non-ACGT = A like SeqAn's Dna).  Returns the number
 * of hits; when out != NULL the first `cap` of them are stored, sorted by (guide, strand, contig, pos).
 * *candidates (optional) receives the number of delegate calls.
 */
long orc_pigeon_search(const orc_pigeon *ix, const char *reads, uint32_t n_reads, uint32_t max_mm, const char *extra_pam,
                       int threads, orc_hit *out, long cap, long *candidates)
{
    if (!ix || max_mm > 8) return -1;
    long total = 0, cand = 0;
    int fwd[3][2] = {{2, 2}, {2, 0}, {0, 0}}; /* GG, GA (:240) */
    int n_pam = 2;
    if (extra_pam && extra_pam[0] && extra_pam[1]) {
        const int a = base_code(extra_pam[0]), b = base_code(extra_pam[1]);
        if (a < 4 && b < 4) {
            fwd[2][0] = a;
            fwd[2][1] = b;
            n_pam = 3;
        }
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    /* per-read results are gathered per thread and merged at the end */
    long *per_read = (long *)calloc((size_t)n_reads + 1, sizeof(long));
    orc_hit **bufs = (orc_hit **)calloc((size_t)n_reads + 1, sizeof(orc_hit *));
    if (!per_read || !bufs) {
        free(per_read);
        free(bufs);
        return -1;
    }
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total, cand)
    for (long g = 0; g < (long)n_reads; ++g) {
        uint8_t fw[RL], rc[RL];
        for (int i = 0; i < RL; ++i) {
            int c = base_code(reads[(size_t)g * RL + i]);
            if (c > 3) c = 0;
            fw[i] = (uint8_t)c;
        }
        for (int i = 0; i < RL; ++i) rc[i] = (uint8_t)(3 - fw[RL - 1 - i]);
        long local_cap = out ? 256 : 0, local_n = 0;
        orc_hit *local = out ? (orc_hit *)malloc((size_t)local_cap * sizeof(orc_hit)) : NULL;
        for (int strand = 0; strand < 2; ++strand) {  /* :291-294: the read, then its reverse complement */
            for (int half = 0; half < 2; ++half) {
                for (;;) {
                    search_state s;
                    memset(&s, 0, sizeof s);
                    s.ix = ix;
                    s.read = strand ? rc : fw;
                    s.guide = (uint32_t)g;
                    s.strand = strand;
                    s.max_mm = (int)max_mm;
                    s.k = (int)max_mm / 2;
                    s.first_half = half == 0;
                    s.n_pam = n_pam;
                    for (int q = 0; q < n_pam; ++q) {
                        if (!strand) {
                            s.pam[q][0] = fwd[q][0];
                            s.pam[q][1] = fwd[q][1];
                        } else { /* reverse complement of the PAM at the window's start (:242-247) */
                            s.pam[q][0] = 3 - fwd[q][1];
                            s.pam[q][1] = 3 - fwd[q][0];
                        }
                    }
                    s.out = local ? local + local_n : NULL;
                    s.cap = local_cap - local_n;
                    enumerate(&s, s.read + (half ? H1 : 0), half ? H2 : H1, 0, s.k, 0, half);
                    if (local && s.n_hits > s.cap) { /* grow and redo this half */
                        local_cap = (local_n + s.n_hits) * 2;
                        local = (orc_hit *)realloc(local, (size_t)local_cap * sizeof(orc_hit));
                        continue;
                    }
                    local_n += s.n_hits;
                    cand += s.candidates;
                    break;
                }
            }
        }
        total += local_n;
        per_read[g] = local_n;
        bufs[g] = local;
    }
    if (out) {
        long at = 0;
        for (uint32_t g = 0; g < n_reads; ++g) {
            if (bufs[g]) {
                qsort(bufs[g], (size_t)per_read[g], sizeof(orc_hit), cmp_hit);
                for (long i = 0; i < per_read[g] && at < cap; ++i) out[at++] = bufs[g][i];
                free(bufs[g]);
            }
        }
    }
    free(per_read);
    free(bufs);
    if (candidates) *candidates = cand;
    return total;
}
