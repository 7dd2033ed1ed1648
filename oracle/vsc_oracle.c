/*
 * vsc_oracle.c - character-level CPU restatement of VARSCOT's read_mapping hot path and of the
 * per-hit scores.  TEST INFRASTRUCTURE ONLY (see vsc_oracle.h for the rules and parity status:
 * search parity is UNPINNED - SeqAn and the reference's own outputs are absent; feature matrix and
 * MIT score are pinned by tests/golden/).
 *
 * Deliberately slow and literal: one char per base, no packing, no threads.  Every function cites
 * the reference lines it follows (relative to /root/reference/VARSCOT_pipeline/).
 */
#include "vsc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RL ORC_READ_LEN

/* SeqAn Dna5 conversion of genome characters: ACGT (any case) kept, everything else N
 * (read_mapping/bidir_index.cpp:36-40 reads the genome as Dna5String). */
static char to_dna5(char c)
{
    switch (c) {
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': return 'T';
    default: return 'N';
    }
}

/* SeqAn Dna conversion of read characters: everything except ACGT becomes A
 * (read_mapping/bidir_mapping.cpp:194 "everything else than ACGT will be converted to A", :256). */
static char to_dna4(char c)
{
    char d = to_dna5(c);
    return d == 'N' ? 'A' : d;
}

static char complement(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default: return 'N';
    }
}

static void revcomp(const char *in, int n, char *out)
{
    for (int i = 0; i < n; ++i)
        out[i] = complement(in[n - 1 - i]);
}

/* bidir_mapping.cpp:240-247: valid PAM lists.  fwd = {GG, GA} (+P), rev = {CC, TC} (+revcomp(P)). */
typedef struct {
    char fwd[3][2];
    char rev[3][2];
    int n;
} pam_sets;

static void make_pams(const char *extra, pam_sets *p)
{
    memcpy(p->fwd[0], "GG", 2);
    memcpy(p->fwd[1], "GA", 2);
    memcpy(p->rev[0], "CC", 2);
    memcpy(p->rev[1], "TC", 2);
    p->n = 2;
    if (extra && extra[0] && extra[1]) {
        char e[2] = { to_dna5(extra[0]), to_dna5(extra[1]) };
        memcpy(p->fwd[2], e, 2);
        revcomp(e, 2, p->rev[2]);
        p->n = 3;
    }
}

/* bidir_mapping.cpp:21-29 */
static int is_valid_pam(const char *two, const char (*valid)[2], int n)
{
    for (int i = 0; i < n; ++i)
        if (two[0] == valid[i][0] && two[1] == valid[i][1])
            return 1;
    return 0;
}

/* ---- growable hit list -------------------------------------------------------------------- */
typedef struct {
    orc_hit *v;
    long n, cap_store; /* stored up to cap_store, counted beyond */
} hit_sink;

static void sink_push(hit_sink *s, orc_hit h)
{
    if (s->n < s->cap_store)
        s->v[s->n] = h;
    s->n++;
}

static uint32_t make_info(int rev, int secondary, unsigned nm, uint32_t mask)
{
    return ((uint32_t)rev << 31) | ((uint32_t)secondary << 30) | ((nm & 31u) << 23) | (mask & 0x7FFFFFu);
}

/* ---- ORC_MODE_PREDICATE: SURVEY.md section 8.1 --------------------------------------------- */
static void search_predicate(char *const *text, const uint32_t *len, uint32_t nc, const char *guides,
                             uint32_t ng, unsigned m, const pam_sets *pams, hit_sink *out)
{
    unsigned k = m / 2; /* bidir_mapping.cpp:129-146 */
    for (uint32_t g = 0; g < ng; ++g) {
        char fwd[RL], rc[RL];
        for (int i = 0; i < RL; ++i)
            fwd[i] = to_dna4(guides[(size_t)g * RL + i]);
        revcomp(fwd, RL, rc);
        for (int rev = 0; rev < 2; ++rev) {
            const char *r = rev ? rc : fwd;
            for (uint32_t c = 0; c < nc; ++c) {
                if (len[c] < RL)
                    continue;
                for (uint32_t p = 0; p + RL <= len[c]; ++p) {
                    const char *w = text[c] + p;
                    /* 2. PAM */
                    if (!rev && !is_valid_pam(w + RL - 2, pams->fwd, pams->n))
                        continue;
                    if (rev && !is_valid_pam(w, pams->rev, pams->n))
                        continue;
                    /* 3. no N, 4. HD <= m */
                    unsigned nm = 0, second = 0;
                    uint32_t mask = 0;
                    int has_n = 0;
                    for (int i = 0; i < RL; ++i) {
                        if (w[i] == 'N')
                            has_n = 1;
                        if (w[i] != r[i]) {
                            nm++;
                            mask |= 1u << i;
                            if (i >= RL / 2)
                                second++;
                        }
                    }
                    if (has_n || nm > m)
                        continue;
                    /* 1. right-edge rule: only the second-half route reports p + 23 == L */
                    if (p + RL == len[c] && second > k)
                        continue;
                    orc_hit h = { g, c, p, make_info(rev, 0, nm, mask) };
                    sink_push(out, h);
                }
            }
        }
    }
}

/* ---- ORC_MODE_REFERENCE_FLOW ---------------------------------------------------------------- */
typedef struct {
    uint32_t key_contig; /* uint16_t-truncated when compat_u16 (bidir_mapping.cpp:13) */
    uint32_t contig, pos;
    unsigned mismatches;
    uint32_t mask;
} record;

typedef struct {
    record *v;
    size_t n, cap;
} record_map; /* std::map<TOccType, BamRecord>, bidir_mapping.cpp:154: kept sorted by (key_contig, pos) */

static long map_find(const record_map *m, uint32_t kc, uint32_t pos, int *found)
{
    size_t lo = 0, hi = m->n;
    while (lo < hi) {
        size_t mid = (lo + hi) / 2;
        const record *r = &m->v[mid];
        if (r->key_contig < kc || (r->key_contig == kc && r->pos < pos))
            lo = mid + 1;
        else
            hi = mid;
    }
    *found = lo < m->n && m->v[lo].key_contig == kc && m->v[lo].pos == pos;
    return (long)lo;
}

static void map_insert_at(record_map *m, long at, record r)
{
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 64;
        m->v = (record *)realloc(m->v, m->cap * sizeof(record));
    }
    memmove(m->v + at + 1, m->v + at, (m->n - (size_t)at) * sizeof(record));
    m->v[at] = r;
    m->n++;
}

/* bidir_mapping.cpp:31-148.  `partial` is the half being searched, `first_half` says which. */
static void search_and_verify(char *const *text, const uint32_t *len, uint32_t nc, const char *full_read,
                              const char *partial, int plen, int reverse_strand, int first_half,
                              record_map *records, unsigned max_mm, const pam_sets *pams, int compat_u16)
{
    unsigned k = max_mm / 2; /* :129-146: find<0, k> with k = 0,0,1,1,2,2,3,3,4 */
    /* find<0,k>(delegate, index, partialRead, HammingDistance()) + getOccurrences (:39): every
     * (contig, offset) at which the half occurs with at most k substitutions.  The index is built
     * over the Dna5 contigs as a StringSet, so occurrences never span contigs and a genome N is a
     * mismatch against every read character. */
    for (uint32_t c = 0; c < nc; ++c) {
        if (len[c] < (uint32_t)plen)
            continue;
        for (uint32_t occ = 0; occ + (uint32_t)plen <= len[c]; ++occ) {
            unsigned e = 0;
            for (int i = 0; i < plen && e <= k; ++i)
                e += text[c][occ + i] != partial[i];
            if (e > k)
                continue;
            /* ---- delegate body, :41-125 ---- */
            unsigned mismatches = 0;
            uint32_t pos = occ;
            if (first_half) { /* :48-53 */
                if (len[c] <= pos + RL)
                    continue;
            } else { /* :54-62 */
                if ((long)pos - (long)(RL - plen) < 0)
                    continue;
                pos -= (uint32_t)(RL - plen);
            }
            uint32_t kc = compat_u16 ? (c & 0xFFFFu) : c;
            int found;
            long at = map_find(records, kc, pos, &found); /* :64-65 */
            if (found)
                continue;
            const char *mapped = text[c] + pos; /* :67 */
            if (!reverse_strand && !is_valid_pam(mapped + RL - 2, pams->fwd, pams->n)) /* :71-72 */
                continue;
            if (reverse_strand && !is_valid_pam(mapped, pams->rev, pams->n)) /* :75-76 */
                continue;
            uint32_t mask = 0;
            for (unsigned i = 0; i < RL && mismatches <= max_mm; ++i) { /* :79-84 */
                if (mapped[i] == 'N')
                    mismatches += max_mm + 1;
                if (full_read[i] != mapped[i]) {
                    mismatches += 1;
                    mask |= 1u << i;
                }
            }
            if (mismatches > max_mm) /* :85-86 */
                continue;
            record r = { kc, c, pos, mismatches, mask };
            map_insert_at(records, at, r); /* :125 */
        }
    }
}

/* bidir_mapping.cpp:150-188 */
static void search_entire_read(char *const *text, const uint32_t *len, uint32_t nc, uint32_t guide,
                               const char *read, int reverse_strand, unsigned max_mm, const pam_sets *pams,
                               int compat_u16, hit_sink *out)
{
    record_map records = { 0, 0, 0 };
    int half = RL / 2; /* :157 length(read)/2 = 11 */
    search_and_verify(text, len, nc, read, read, half, reverse_strand, 1, &records, max_mm, pams, compat_u16);
    search_and_verify(text, len, nc, read, read + half, RL - half, reverse_strand, 0, &records, max_mm, pams,
                      compat_u16); /* :161-162 */
    if (records.n == 0) { /* :164-165 */
        free(records.v);
        return;
    }
    size_t best = 0; /* :167 */
    for (size_t it = 1; it < records.n; ++it) { /* :170-185 */
        if (records.v[it].mismatches >= records.v[best].mismatches) {
            orc_hit h = { guide, records.v[it].contig, records.v[it].pos,
                          make_info(reverse_strand, 1, records.v[it].mismatches, records.v[it].mask) };
            sink_push(out, h);
        } else {
            orc_hit h = { guide, records.v[best].contig, records.v[best].pos,
                          make_info(reverse_strand, 1, records.v[best].mismatches, records.v[best].mask) };
            sink_push(out, h);
            best = it;
        }
    }
    orc_hit h = { guide, records.v[best].contig, records.v[best].pos,
                  make_info(reverse_strand, 0, records.v[best].mismatches, records.v[best].mask) };
    sink_push(out, h); /* :187 */
    free(records.v);
}

static char **normalise_genome(const char *const *contigs, const uint32_t *len, uint32_t nc)
{
    char **text = (char **)calloc(nc ? nc : 1, sizeof(char *));
    for (uint32_t c = 0; c < nc; ++c) {
        text[c] = (char *)malloc(len[c] ? len[c] : 1);
        for (uint32_t i = 0; i < len[c]; ++i)
            text[c][i] = to_dna5(contigs[c][i]);
    }
    return text;
}

static void free_genome(char **text, uint32_t nc)
{
    for (uint32_t c = 0; c < nc; ++c)
        free(text[c]);
    free(text);
}

long orc_search(const char *const *contigs, const uint32_t *contig_len, uint32_t n_contigs, const char *guides,
                uint32_t n_guides, uint32_t max_mm, const char *extra_pam, int mode, int compat_u16,
                orc_hit *out, long cap)
{
    if (max_mm > 8) /* bidir_mapping.cpp:234-238 */
        return -1;
    pam_sets pams;
    make_pams(extra_pam, &pams);
    char **text = normalise_genome(contigs, contig_len, n_contigs);
    hit_sink sink = { out, 0, out ? cap : 0 };
    if (mode == ORC_MODE_PREDICATE) {
        search_predicate(text, contig_len, n_contigs, guides, n_guides, max_mm, &pams, &sink);
    } else {
        /* bidir_mapping.cpp:285-295: reads in input order, forward then reverse complement */
        for (uint32_t g = 0; g < n_guides; ++g) {
            char read[RL], rc[RL];
            for (int i = 0; i < RL; ++i)
                read[i] = to_dna4(guides[(size_t)g * RL + i]);
            search_entire_read(text, contig_len, n_contigs, g, read, 0, max_mm, &pams, compat_u16, &sink);
            revcomp(read, RL, rc); /* :293 */
            search_entire_read(text, contig_len, n_contigs, g, rc, 1, max_mm, &pams, compat_u16, &sink);
        }
    }
    free_genome(text, n_contigs);
    return sink.n;
}

/* ---- MD string, SAM text ---------------------------------------------------------------------- */
void orc_md_string(const char *window, const char *read, int md_style, char *out)
{
    /* bidir_mapping.cpp:114-119: getMDString(md, row0 = mappedRegion, row1 = fullRead); SeqAn's
     * implementation is not available - both plausible conventions are provided. */
    int run = 0, n = 0;
    for (int i = 0; i < RL; ++i) {
        if (window[i] == read[i]) {
            run++;
        } else {
            if (run > 0 || md_style == 0)
                n += sprintf(out + n, "%d", run);
            out[n++] = window[i];
            run = 0;
        }
    }
    if (run > 0 || md_style == 0)
        n += sprintf(out + n, "%d", run);
    out[n] = 0;
}

int orc_md_positions(const char *md, int *out)
{
    /* filter_output_bam.h:334-348: while (is >> num >> base) { pos += num + 1; push(pos - 1); } */
    int n = 0;
    unsigned pos = 0;
    const char *p = md;
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == '\n')
            p++;
        if (*p < '0' || *p > '9')
            break; /* operator>>(unsigned) fails */
        unsigned num = 0;
        while (*p >= '0' && *p <= '9')
            num = num * 10 + (unsigned)(*p++ - '0');
        while (*p == ' ' || *p == '\t' || *p == '\n')
            p++;
        if (!*p)
            break; /* operator>>(char) fails at end of string */
        p++;
        pos += num + 1;
        if (n < 24)
            out[n] = (int)pos - 1;
        n++;
    }
    if (n == 0) {
        out[0] = -1;
        return 1;
    }
    return n;
}

typedef struct {
    char *s;
    size_t n, cap;
} strbuf;

static void sb_add(strbuf *b, const char *s, size_t n)
{
    if (b->n + n + 1 > b->cap) {
        b->cap = (b->n + n + 1) * 2;
        b->s = (char *)realloc(b->s, b->cap);
    }
    memcpy(b->s + b->n, s, n);
    b->n += n;
    b->s[b->n] = 0;
}

char *orc_search_sam(const char *const *contigs, const uint32_t *contig_len, const char *const *contig_names,
                     uint32_t n_contigs, const char *guides, const char *const *guide_names, uint32_t n_guides,
                     uint32_t max_mm, const char *extra_pam, int md_style)
{
    long n = orc_search(contigs, contig_len, n_contigs, guides, n_guides, max_mm, extra_pam,
                        ORC_MODE_REFERENCE_FLOW, 0, NULL, 0);
    if (n < 0)
        return NULL;
    orc_hit *hits = (orc_hit *)malloc((n ? n : 1) * sizeof(orc_hit));
    orc_search(contigs, contig_len, n_contigs, guides, n_guides, max_mm, extra_pam, ORC_MODE_REFERENCE_FLOW, 0,
               hits, n);
    strbuf sb = { 0, 0, 0 };
    sb_add(&sb, "", 0);
    for (long i = 0; i < n; ++i) {
        const orc_hit *h = &hits[i];
        int rev = ORC_INFO_STRAND(h->info);
        char seq[RL + 1], rc[RL + 1], window[RL + 1], md[64];
        for (int j = 0; j < RL; ++j) {
            seq[j] = to_dna4(guides[(size_t)h->guide * RL + j]);
            window[j] = to_dna5(contigs[h->contig][h->pos + j]);
        }
        seq[RL] = rc[RL] = window[RL] = 0;
        revcomp(seq, RL, rc);
        /* :106-108: SEQ = fullRead, reverse-complemented again for '-' hits = the original guide.
         * MD is computed against fullRead (:117-119), i.e. revcomp(guide) for '-'. */
        orc_md_string(window, rev ? rc : seq, md_style, md);
        unsigned flag = (rev ? 16u : 0u) | (ORC_INFO_SECONDARY(h->info) ? 256u : 0u);
        char line[512];
        int len = snprintf(line, sizeof line, "%s\t%u\t%s\t%u\t255\t23M\t*\t0\t0\t%s\tIIIIIIIIIIIIIIIIIIIIIII\tNM:i:%u\tMD:Z:%s\n",
                           guide_names[h->guide], flag, contig_names[h->contig], h->pos + 1, seq,
                           ORC_INFO_NM(h->info), md);
        sb_add(&sb, line, (size_t)len);
    }
    free(hits);
    return sb.s;
}

void orc_free(void *p) { free(p); }

/* ---- MIT score: variant_processing/mit_score.h:12-68 ------------------------------------------ */
double orc_mit_score(const int *mismatchPos, int n, int *ub)
{
    double s, s1, s2, s3 = 0, avgDist;
    unsigned nm;
    if (ub)
        *ub = 0;
    if (n == 1 && mismatchPos[0] == -1) { /* :19-22 */
        nm = 0;
    } else {
        if (mismatchPos[n - 1] < 20) /* :26-33 exclude (one) mismatch in the PAM */
            nm = (unsigned)n;
        else
            nm = (unsigned)n - 1;
        s3 = (double)1 / (double)pow((double)nm, 2); /* :35 */
    }
    if (nm == 0) /* :38-41 */
        return 100;
    static const double matrixM[20] = { 0, 0, 0.014, 0, 0, 0.395, 0.317, 0, 0.389, 0.079,
                                        0.445, 0.508, 0.613, 0.851, 0.732, 0.828, 0.615, 0.804, 0.685, 0.583 };
    s1 = 1;
    int distSum = 0; /* std::accumulate(dist.begin(), dist.end(), 0) - an int accumulator, :63 */
    unsigned distCount = 0;
    for (unsigned i = 0; i < nm; i++) { /* :48-55 */
        double w;
        if (mismatchPos[i] >= 0 && mismatchPos[i] < 20) {
            w = matrixM[mismatchPos[i]];
        } else { /* reference reads out of bounds here: defined as weight 0 and flagged */
            w = 0;
            if (ub)
                *ub = 1;
        }
        s1 *= (1 - w);
        if (i > 0) {
            distSum += (int)(unsigned)(mismatchPos[i] - mismatchPos[i - 1]);
            distCount++;
        }
    }
    if (nm < 2) { /* :57-60 */
        s2 = 1;
    } else {
        avgDist = (double)distSum / (double)distCount; /* :63 */
        s2 = 1 / (((19 - avgDist) / 19) * 4 + 1);      /* :64 */
    }
    s = s1 * s2 * s3 * 100; /* :66 */
    return s;
}

/* ---- feature matrix: variant_processing/feature_matrix.h:25-126 -------------------------------- */
static int base_index(char c) /* A,C,G,T -> 0..3; everything else is treated as A (:80-82) */
{
    switch (c) {
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 0;
    }
}

void orc_feature_row(const char *on, const char *off, uint32_t *f)
{
    memset(f, 0, 442 * sizeof(uint32_t));
    /* :45-46 AC,AG,AT,CA,CG,CT,GA,GC,GT,TA,TC,TG -> 0..11 */
    static const int mismatchType[4][4] = { { -1, 0, 1, 2 }, { 3, -1, 4, 5 }, { 6, 7, -1, 8 }, { 9, 10, 11, -1 } };
    int precMismatch = 0;
    for (unsigned i = 0; i < RL - 2; ++i) { /* :53 */
        int b = base_index(off[i]);
        if (i < 19) { /* :56-61 */
            int pair = b * 4 + base_index(off[i + 1]);
            f[120 + i * 16 + pair] = 1;
            f[424 + pair]++;
        }
        f[36 + i * 4 + b] = 1; /* :64-83 */
        if (on[i] != off[i]) { /* :86 */
            f[0]++;       /* :89 */
            f[i + 1] = 1; /* :92 */
            if (i > 7 && i < 20) /* :94-98 */
                f[441]++;
            if (precMismatch) /* :100-103 */
                f[440]++;
            precMismatch = 1;
            int a = base_index(on[i]);
            /* :47 transitions AG, CT, GA, TC */
            if ((a == 0 && b == 2) || (a == 1 && b == 3) || (a == 2 && b == 0) || (a == 3 && b == 1))
                f[34]++;
            else
                f[35]++;
            /* :119 features[22 + mismatchTypes[currMismatch]] = 1; std::map::operator[] yields 0 for a
             * pair it does not hold (only reachable with a non-ACGT on-target letter) */
            f[22 + (mismatchType[a][b] >= 0 ? mismatchType[a][b] : 0)] = 1;
        } else {
            precMismatch = 0; /* :123 */
        }
    }
}
