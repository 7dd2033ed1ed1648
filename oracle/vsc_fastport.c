/*
 * vsc_fastport.c - bit-parallel, OpenMP port of the search predicate (which
 * restates read_mapping/bidir_mapping.cpp:31-188) for inputs the character-level restatement in
 * vsc_oracle.c would need minutes for, and for bench.py's cpu_baseline leg ("kind": "port").
 *
 * TEST INFRASTRUCTURE ONLY - see vsc_oracle.h.  Search parity is UNPINNED (no SeqAn, no reference
 * outputs); tests/test_oracle.py checks this file against vsc_oracle.c on small genomes.
 *
 * Method: every contig is walked once with a rolling 46-bit window (2 bits per base, base i of the
 * window in bits 2i..2i+1, A=0 C=1 G=2 T=3) and a count of N characters inside the window; windows
 * with a valid PAM (bidir_mapping.cpp:71-76) and no N (:81-82) are compared against every read
 * (forward reads for '+', reverse-complemented reads for '-', :291-294) with XOR + popcount (:79-86).
 */
#include "vsc_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RL ORC_READ_LEN
#define MASK46 ((1ULL << 46) - 1)
#define LOW_OF_PAIRS 0x155555555555ULL /* 23 pairs */

static int code_of(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

typedef struct {
    orc_hit *v;
    long n, cap;
} hitvec;

static void hv_push(hitvec *h, orc_hit x)
{
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 1024;
        h->v = (orc_hit *)realloc(h->v, (size_t)h->cap * sizeof(orc_hit));
    }
    h->v[h->n++] = x;
}

static int cmp_hit(const void *a, const void *b)
{
    const orc_hit *x = (const orc_hit *)a, *y = (const orc_hit *)b;
    if (x->guide != y->guide)
        return x->guide < y->guide ? -1 : 1;
    unsigned sx = x->info >> 31, sy = y->info >> 31;
    if (sx != sy)
        return sx < sy ? -1 : 1;
    if (x->contig != y->contig)
        return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos)
        return x->pos < y->pos ? -1 : 1;
    return 0;
}

/* pair-fold: one bit per mismatching base, at bit 2i */
static inline uint64_t fold(uint64_t x) { return (x | (x >> 1)) & LOW_OF_PAIRS; }

static inline uint32_t compact_mask(uint64_t t)
{
    uint32_t m = 0;
    for (int i = 0; i < RL; ++i)
        m |= (uint32_t)((t >> (2 * i)) & 1u) << i;
    return m;
}

typedef struct {
    uint32_t contig, begin, end; /* window starts [begin, end) */
} chunk;

static long run(const char *const *contigs, const uint32_t *len, uint32_t nc, const char *guides, uint32_t ng,
                uint32_t m, const char *extra_pam, int threads, orc_hit *out, long cap, int count_only, long *sites_out)
{
    if (m > 8)
        return -1;
    unsigned k = m / 2;
    /* PAM tables indexed by (first base code) * 4 + second base code */
    unsigned char fwd_ok[16] = { 0 }, rev_ok[16] = { 0 };
    fwd_ok[2 * 4 + 2] = fwd_ok[2 * 4 + 0] = 1; /* GG GA */
    rev_ok[1 * 4 + 1] = rev_ok[3 * 4 + 1] = 1; /* CC TC */
    if (extra_pam && extra_pam[0] && extra_pam[1]) {
        int a = code_of(extra_pam[0]), b = code_of(extra_pam[1]);
        if (a < 4 && b < 4) {
            fwd_ok[a * 4 + b] = 1;
            rev_ok[(3 - b) * 4 + (3 - a)] = 1; /* reverse complement */
        }
    }
    uint64_t *rf = (uint64_t *)malloc((ng ? ng : 1) * sizeof(uint64_t));
    uint64_t *rr = (uint64_t *)malloc((ng ? ng : 1) * sizeof(uint64_t));
    for (uint32_t g = 0; g < ng; ++g) {
        uint64_t f = 0, r = 0;
        for (int i = 0; i < RL; ++i) {
            int c = code_of(guides[(size_t)g * RL + i]);
            if (c == 4)
                c = 0; /* Dna conversion: non-ACGT -> A */
            f |= (uint64_t)c << (2 * i);
            r |= (uint64_t)(3 - c) << (2 * (RL - 1 - i));
        }
        rf[g] = f;
        rr[g] = r;
    }
    /* chunks of window starts */
    const uint32_t CH = 1u << 20;
    size_t nchunks = 0;
    for (uint32_t c = 0; c < nc; ++c)
        if (len[c] >= RL)
            nchunks += (len[c] - RL + 1 + CH - 1) / CH;
    chunk *chunks = (chunk *)malloc((nchunks ? nchunks : 1) * sizeof(chunk));
    size_t ci = 0;
    for (uint32_t c = 0; c < nc; ++c) {
        if (len[c] < RL)
            continue;
        uint32_t nw = len[c] - RL + 1;
        for (uint32_t b = 0; b < nw; b += CH) {
            chunks[ci].contig = c;
            chunks[ci].begin = b;
            chunks[ci].end = b + CH < nw ? b + CH : nw;
            ci++;
        }
    }
    int nt = 1;
#ifdef _OPENMP
    nt = threads > 0 ? threads : omp_get_max_threads();
#else
    (void)threads;
#endif
    hitvec *hv = (hitvec *)calloc((size_t)nt, sizeof(hitvec));
    long *cnt = (long *)calloc((size_t)nt * 2, sizeof(long));

#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
    for (long q = 0; q < (long)nchunks; ++q) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const chunk *ch = &chunks[q];
        const char *s = contigs[ch->contig];
        uint32_t L = len[ch->contig];
        uint64_t w = 0;
        int nn = 0; /* N characters inside the current window */
        /* prime with the first RL-1 bases */
        for (uint32_t i = 0; i < RL - 1; ++i) {
            int c = code_of(s[ch->begin + i]);
            nn += c == 4;
            w |= (uint64_t)(c & 3) << (2 * i);
        }
        long local_hits = 0, local_sites = 0;
        for (uint32_t p = ch->begin; p < ch->end; ++p) {
            int c = code_of(s[p + RL - 1]);
            nn += c == 4;
            w |= (uint64_t)(c & 3) << (2 * (RL - 1));
            if (nn == 0) {
                int f_ok = fwd_ok[((w >> 42) & 3) * 4 + ((w >> 44) & 3)];
                int r_ok = rev_ok[(w & 3) * 4 + ((w >> 2) & 3)];
                int at_edge = p + RL == L;
                for (int rev = 0; rev < 2; ++rev) {
                    if (!(rev ? r_ok : f_ok))
                        continue;
                    local_sites++;
                    const uint64_t *reads = rev ? rr : rf;
                    for (uint32_t g = 0; g < ng; ++g) {
                        uint64_t t = fold(w ^ reads[g]);
                        unsigned nm = (unsigned)__builtin_popcountll(t);
                        if (nm > m)
                            continue;
                        /* right-edge rule (bidir_mapping.cpp:51-52): only the second-half route
                         * can report a window that ends exactly at the contig end */
                        if (at_edge && (unsigned)__builtin_popcountll(t >> (2 * (RL / 2))) > k)
                            continue;
                        local_hits++;
                        if (!count_only) {
                            orc_hit h = { g, ch->contig, p,
                                          ((uint32_t)rev << 31) | (nm << 23) | compact_mask(t) };
                            hv_push(&hv[tid], h);
                        }
                    }
                }
            }
            /* slide */
            nn -= code_of(s[p]) == 4;
            w >>= 2;
        }
        cnt[tid * 2] += local_hits;
        cnt[tid * 2 + 1] += local_sites;
    }
    long total = 0, sites = 0;
    for (int t = 0; t < nt; ++t) {
        total += cnt[t * 2];
        sites += cnt[t * 2 + 1];
    }
    if (sites_out)
        *sites_out = sites;
    if (!count_only) {
        orc_hit *all = (orc_hit *)malloc((size_t)(total ? total : 1) * sizeof(orc_hit));
        long o = 0;
        for (int t = 0; t < nt; ++t) {
            if (hv[t].n)
                memcpy(all + o, hv[t].v, (size_t)hv[t].n * sizeof(orc_hit));
            o += hv[t].n;
        }
        qsort(all, (size_t)total, sizeof(orc_hit), cmp_hit);
        long ncopy = total < cap ? total : cap;
        if (out && ncopy > 0)
            memcpy(out, all, (size_t)ncopy * sizeof(orc_hit));
        free(all);
    }
    for (int t = 0; t < nt; ++t)
        free(hv[t].v);
    free(hv);
    free(cnt);
    free(chunks);
    free(rf);
    free(rr);
    return total;
}

long orc_search_fast(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs, const char *guides,
                     uint32_t n_guides, uint32_t max_mm, const char *extra_pam, int threads, orc_hit *out, long cap)
{
    return run(contigs, contig_len, n_contigs, guides, n_guides, max_mm, extra_pam, threads, out, cap, 0, NULL);
}

long orc_count_fast(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs, const char *guides,
                    uint32_t n_guides, uint32_t max_mm, const char *extra_pam, int threads, long *sites)
{
    return run(contigs, contig_len, n_contigs, guides, n_guides, max_mm, extra_pam, threads, NULL, 0, 1, sites);
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
