// vsc_pack.cpp - host-side packing of sequences into the plane layout of include/varscot_hip.h.
// Pure C++ (no device code): the counterpart of the reference's in-memory genome / read
// containers (StringSet<Dna5String> in read_mapping/bidir_index.cpp:36-40, StringSet<DnaString> in
// read_mapping/bidir_mapping.cpp:256,264).
#include <cstring>

#include "varscot_hip.h"

namespace {

// SeqAn Dna5 conversion: ACGT (any case) -> 0..3, everything else -> 4 (N)
inline int code_of(char c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

}  // namespace

extern "C" {

uint64_t vsc_layout_contigs(const uint32_t *contig_len, uint32_t n_contigs, vsc_contig *out_table)
{
    uint64_t pos = 0;
    for (uint32_t c = 0; c < n_contigs; ++c) {
        if (out_table) {
            out_table[c].offset = pos;
            out_table[c].length = contig_len[c];
            out_table[c].reserved = 0;
        }
        pos += (uint64_t)contig_len[c] + 1;  // one N separator after every contig
    }
    return (pos + 31) / 32;
}

void vsc_planes_init(uint32_t *hi, uint32_t *lo, uint32_t *nmask, uint64_t n_words)
{
    std::memset(hi, 0, n_words * sizeof(uint32_t));
    std::memset(lo, 0, n_words * sizeof(uint32_t));
    std::memset(nmask, 0xFF, n_words * sizeof(uint32_t));
}

void vsc_pack_bases(const char *seq, uint64_t n, uint64_t dst_pos, uint32_t *hi, uint32_t *lo, uint32_t *nmask)
{
    static const struct Lut {
        uint8_t c[256];
        Lut()
        {
            for (int i = 0; i < 256; ++i) c[i] = (uint8_t)code_of((char)i);
        }
    } lut;
    auto put = [&](uint64_t p, int c) {
        const uint64_t w = p >> 5;
        const uint32_t bit = 1u << (p & 31);
        if (c == 4) {
            hi[w] &= ~bit;
            lo[w] &= ~bit;
            nmask[w] |= bit;
        } else {
            hi[w] = (c & 2) ? (hi[w] | bit) : (hi[w] & ~bit);
            lo[w] = (c & 1) ? (lo[w] | bit) : (lo[w] & ~bit);
            nmask[w] &= ~bit;
        }
    };
    uint64_t i = 0;
    while (i < n && ((dst_pos + i) & 31)) {  // up to the first word boundary
        put(dst_pos + i, lut.c[(uint8_t)seq[i]]);
        ++i;
    }
    for (; i + 32 <= n; i += 32) {  // whole words: 32 bases at a time
        uint32_t h = 0, l = 0, m = 0;
        const uint8_t *s = (const uint8_t *)seq + i;
        for (int b = 0; b < 32; ++b) {
            const uint32_t c = lut.c[s[b]];
            h |= ((c >> 1) & 1u) << b;
            l |= (c & 1u) << b;
            m |= (c >> 2) << b;
        }
        const uint64_t w = (dst_pos + i) >> 5;
        hi[w] = h & ~m;
        lo[w] = l & ~m;
        nmask[w] = m;
    }
    for (; i < n; ++i) put(dst_pos + i, lut.c[(uint8_t)seq[i]]);
}

void vsc_unpack_bases(const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t src_pos, uint64_t n,
                      char *out)
{
    static const char kLetters[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t p = src_pos + i;
        const uint64_t w = p >> 5;
        const unsigned b = (unsigned)(p & 31);
        if ((nmask[w] >> b) & 1u)
            out[i] = 'N';
        else
            out[i] = kLetters[(((hi[w] >> b) & 1u) << 1) | ((lo[w] >> b) & 1u)];
    }
}

uint64_t vsc_pack_guide(const char *seq23)
{
    uint64_t g = 0;
    for (int i = 0; i < VSC_READ_LEN; ++i) {
        int c = code_of(seq23[i]);
        if (c == 4) c = 0;  // SeqAn Dna conversion of reads: non-ACGT -> A
        g |= (uint64_t)c << (2 * i);
    }
    return g;
}

void vsc_unpack_features(const uint32_t *packed, uint64_t n, uint8_t *features)
{
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t *w = packed + r * 16;
        uint8_t *f = features + r * VSC_N_FEATURES;
        std::memset(f, 0, VSC_N_FEATURES);
        f[0] = (uint8_t)((w[0] >> 21) & 31u);
        for (int i = 0; i < 21; ++i) f[1 + i] = (uint8_t)((w[0] >> i) & 1u);
        for (int i = 0; i < 12; ++i) f[22 + i] = (uint8_t)((w[1] >> i) & 1u);
        f[34] = (uint8_t)((w[1] >> 12) & 31u);
        f[35] = (uint8_t)((w[1] >> 17) & 31u);
        for (int b = 0; b < 84; ++b) f[36 + b] = (uint8_t)((w[2 + (b >> 5)] >> (b & 31)) & 1u);
        for (int b = 0; b < 304; ++b) {
            const uint8_t v = (uint8_t)((w[5 + (b >> 5)] >> (b & 31)) & 1u);
            f[120 + b] = v;
            f[424 + (b & 15)] += v;  // dinucleotide counts = column sums of the one-hots
        }
        f[440] = (uint8_t)((w[0] >> 26) & 31u);
        f[441] = (uint8_t)((w[1] >> 22) & 15u);
    }
}

void vsc_sam_order(const vsc_hit *hits, uint64_t n, uint64_t *order, uint8_t *secondary)
{
    // read_mapping/bidir_mapping.cpp:167-187, per (read, strand) block of the ascending result:
    // the current best (fewest mismatches, first wins ties) is held back; a record that is not
    // better is written at once with the secondary flag, a better one displaces the held record,
    // which is then written with the secondary flag; the last held record is written unflagged.
    uint64_t i = 0, o = 0;
    while (i < n) {
        uint64_t j = i + 1;
        while (j < n && hits[j].guide == hits[i].guide && VSC_HIT_STRAND(hits[j].info) == VSC_HIT_STRAND(hits[i].info))
            ++j;
        uint64_t best = i;
        for (uint64_t it = i + 1; it < j; ++it) {
            if (VSC_HIT_NM(hits[it].info) >= VSC_HIT_NM(hits[best].info)) {
                order[o] = it;
            } else {
                order[o] = best;
                best = it;
            }
            secondary[o++] = 1;
        }
        order[o] = best;
        secondary[o++] = 0;
        i = j;
    }
}

}  // extern "C"
