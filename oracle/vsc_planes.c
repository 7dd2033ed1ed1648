/*
 * vsc_planes.c - the packed-plane genome layout of include/varscot_hip.h read back into text, for the
 * parity tests that run the oracle on the bench's synthetic genomes (those are generated directly in
 * packed form: there is no FASTA to hand to the oracle).
 *
 * TEST INFRASTRUCTURE ONLY - see vsc_oracle.h.  Deliberately independent of the product's own
 * vsc_unpack_bases: the layout is restated from the header (bit b of 32-bit word w <-> global position
 * 32 w + b; hi:lo = A 00, C 01, G 10, T 11; nmask 1 = N / separator / padding), which is the form in
 * which the replacement keeps what read_mapping/bidir_index.cpp:36-47 keeps as a StringSet<Dna5String>.
 */
#include "vsc_oracle.h"

void orc_planes_to_text(const uint32_t *hi, const uint32_t *lo, const uint32_t *nmask, uint64_t pos, uint64_t n, char *out)
{
    static const char letters[4] = { 'A', 'C', 'G', 'T' };
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        uint64_t p = pos + (uint64_t)i;
        uint64_t w = p >> 5;
        unsigned b = (unsigned)(p & 31);
        if ((nmask[w] >> b) & 1u)
            out[i] = 'N';
        else
            out[i] = letters[(((hi[w] >> b) & 1u) << 1) | ((lo[w] >> b) & 1u)];
    }
}
