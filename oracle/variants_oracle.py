"""Pure-Python restatement of VARSCOT's alt-allele expansion (row R8): VCF record -> variants
(variant_processing/process_vcf.h:54-269), overlap sweep (overlap_sequences.h:35-240) and window
sequences / FASTA ids (write_fasta.h:30-470), as `vcf_loader` chains them (vcf_loader.cpp:40-68).

TEST INFRASTRUCTURE ONLY (see oracle/vsc_oracle.h).  Small inputs only - plain loops, one Python
object per variant.  PARITY UNPINNED: the reference has no tests or stored outputs for this stage and
its VCF/FAI/FASTA I/O is SeqAn 2.4.0rc2 (not under /root/reference); the restatement follows the
source line by line.  Where the reference has undefined behaviour the definition chosen here is the
one DESIGN.md records:
  * maxDeletion[-1] (overlap_sequences.h:109,115) reads 0;
  * allVariants[...][j].pos with j past the record's variant count (:101,124) = the record's pos
    (all variants of a record share it);
  * variants[1] = vs after resize(1) (process_vcf.h:147-151) keeps that second-allele variant as
    the record's only variant;
  * std::sort (overlap_sequences.h:233) is taken as stable.
All file:line citations are relative to /root/reference/VARSCOT_pipeline/variant_processing/.
"""
import re


class Variant:
    __slots__ = ("ref", "alt", "chr", "pos", "start", "end", "variantType", "allele")

    def __init__(self):
        self.ref = self.alt = ""
        self.chr = self.pos = self.start = self.end = 0
        self.variantType = 0
        self.allele = 0

    def copy(self):
        v = Variant()
        for k in self.__slots__:
            setattr(v, k, getattr(self, k))
        return v


U32 = 1 << 32


def dna5(s):
    """SeqAn Dna5String conversion of REF / ALT text."""
    return "".join(c.upper() if c.upper() in "ACGT" else "N" for c in s)


def _parse_int(s, i):
    """istream >> int at offset i: skips blanks, optional sign, digits.  Returns (value|None, next)."""
    m = re.match(r"\s*([+-]?\d+)", s[i:])
    if not m:
        return None, i
    return int(m.group(1)), i + m.end()


def process_record(fields, sample_index):
    """process_vcf.h:54-209.  fields = the tab-separated columns of one VCF line.
    Returns [] or the record's variants (chr is filled by the caller)."""
    vs = Variant()
    vs.pos = int(fields[1]) - 1  # record.beginPos, :58
    vs.ref = dna5(fields[3])
    samples = fields[9:]
    if sample_index >= len(samples):  # :61-64
        raise IndexError("ERROR: Sample index out of range.")
    geno = samples[sample_index].split(":")
    fmt = fields[8].split(":")
    position_gt = fmt.index("GT") if "GT" in fmt else 0  # :73-83 (uninitialised if absent; 0 here)
    alts = fields[4].split(",")  # :90-91
    text = geno[position_gt] if position_gt < len(geno) else ""
    phased = True
    first, nxt = _parse_int(text, 0)  # :95
    if first is None or first < 0 or first > len(alts):  # int vs unsigned comparison: negatives fail
        return []
    second = None
    m = re.match(r"\s*(\S)", text[nxt:])  # is >> sep
    if m:
        sep = m.group(1)
        second, _ = _parse_int(text, nxt + m.end())
        if second is not None and 0 <= second <= len(alts):
            if sep == "/":
                phased = False
        else:
            second = None
    if second is None:
        second = first  # :104-108 haploid call
    variants = []
    if first == 0 and second == 0:  # :116-120
        return []
    elif first > 0 and second > 0 and first != second:  # :121-157
        a1, a2 = alts[first - 1], alts[second - 1]
        if a1 != "." and a2 != ".":
            v0 = vs.copy(); v0.allele = 0; v0.alt = dna5(a1)
            v1 = vs.copy(); v1.allele = 1; v1.alt = dna5(a2)
            variants = [v0, v1]
        elif a1 != ".":
            v0 = vs.copy(); v0.allele = 0; v0.alt = dna5(a1)
            variants = [v0]
        elif a2 != ".":
            v0 = vs.copy(); v0.allele = 1; v0.alt = dna5(a2)  # reference-UB, see module docstring
            variants = [v0]
        else:
            return []
    else:  # :158-186
        if alts[0] == ".":
            return []
        v0 = vs.copy()
        if first == 0:
            v0.allele = 1; v0.alt = dna5(alts[second - 1])
        elif second == 0:
            v0.allele = 0; v0.alt = dna5(alts[first - 1])
        else:
            v0.allele = 2; v0.alt = dna5(alts[first - 1])
        variants = [v0]
    for v in variants:  # :189-208
        if not phased and first != second:
            v.allele = -1
        if len(v.ref) > len(v.alt):
            v.variantType = 2
        elif len(v.ref) == len(v.alt):
            v.variantType = 0
        else:
            v.variantType = 1
    return variants


def process_vcf_text(text, sample_index):
    """process_vcf.h:226-269.  Returns (allVariants, chrTable).  Contig ids come from ##contig header
    lines first, then in order of first appearance (SeqAn's VcfIOContext name store)."""
    chr_table, all_variants = [], []
    for line in text.splitlines():
        if line.startswith("##"):
            m = re.match(r"##contig=<.*?ID=([^,>]+)", line)
            if m and m.group(1) not in chr_table:
                chr_table.append(m.group(1))
            continue
        if line.startswith("#") or not line.strip():
            continue
        f = line.rstrip("\n").split("\t")
        if f[0] not in chr_table:
            chr_table.append(f[0])
        vs = process_record(f, sample_index)
        for v in vs:
            v.chr = chr_table.index(f[0])
        if vs:
            all_variants.append(vs)
    return all_variants, chr_table


def find_max_overlap(all_variants, sorted_index, seq_length):
    """overlap_sequences.h:35-162.  Returns (overlapRegions, indexCenterVariants); sets start / end."""
    n = len(sorted_index)
    max_del = [0] * n
    for i in range(n):  # :41-52
        for v in all_variants[sorted_index[i]]:
            if v.variantType == 2 and len(v.ref) - len(v.alt) > max_del[i]:
                max_del[i] = len(v.ref) - len(v.alt)

    def md(k):  # maxDeletion[k]; index -1 reads 0
        return max_del[k] if 0 <= k < n else 0

    def pos(k):
        return all_variants[sorted_index[k]][0].pos

    regions, centers = [], []
    r1 = r2 = 0
    for i in range(n):
        if r2 > i:  # :68
            index_right = r2
            w_right = seq_length + max_del[i]  # :77
            if index_right < n:  # :78-84
                for d in range(i + 1, index_right + 1):
                    w_right += max_del[d]
            while index_right < n and (pos(index_right) - pos(i)) % U32 < w_right:  # :86-94
                w_right += max_del[index_right]
                index_right += 1
            if index_right == r2:  # :97-104
                for v in all_variants[centers[-1]]:
                    v.end = (pos(i) + w_right) % U32
                continue
            r2 = index_right  # :105
            index_left = i - 1  # :108
            w_left = seq_length + md(index_left)
            while index_left >= 0 and (pos(i) - pos(index_left)) % U32 < w_left:  # :111-116
                index_left -= 1
                w_left += md(index_left)
            if index_left + 1 == r1:  # :120-128
                for v in all_variants[centers[-1]]:
                    v.end = (pos(i) + w_right) % U32
                regions[-1] = (regions[-1][0], index_right)
                continue
            r1 = index_left + 1  # :129
        else:  # :131-151
            w_right = seq_length + max_del[i]
            index_right = i + 1
            while index_right < n and (pos(index_right) - pos(i)) % U32 < w_right:
                w_right += max_del[index_right]
                index_right += 1
            r2 = index_right
            w_left = seq_length
            r1 = i
        regions.append((r1, r2))  # :152-153
        centers.append(sorted_index[i])
        for v in all_variants[sorted_index[i]]:  # :156-160 (unsigned arithmetic)
            v.start = (v.pos - w_left + 1) % U32
            v.end = (v.pos + w_right) % U32
    return regions, centers


def get_variant_overlap_ranges(all_variants, chr_number, seq_length):
    """overlap_sequences.h:183-240."""
    sorted_index_all = [[] for _ in range(chr_number)]
    for i, vs in enumerate(all_variants):  # :215-218
        sorted_index_all[vs[0].chr].append(i)
    all_regions, all_centers = [], []
    for c in range(chr_number):  # :230-239
        sorted_index_all[c].sort(key=lambda v: all_variants[v][0].pos)  # stable
        r, ce = find_max_overlap(all_variants, sorted_index_all[c], seq_length)
        all_regions.append(r)
        all_centers.append(ce)
    return all_regions, all_centers, sorted_index_all


def get_fasta_id(all_variants, sorted_index, first, index_center, combination, chr_name):
    """write_fasta.h:30-65."""
    parts = [chr_name, "_", str(all_variants[index_center][0].start), "_"]
    if all(c == -1 for c in combination):
        parts.append("REF")
    else:
        parts.append("ALT")
        for i, c in enumerate(combination):
            if c != -1:
                v = all_variants[sorted_index[first + i]][c]
                parts += ["_", str(v.pos), "_", v.ref, "_", v.alt]
    return "".join(parts)


def all_combinations(all_variants, sorted_index, first, last, index_center, chr_name):
    """write_fasta.h:88-229.  Returns (altCombinations, fastaIDs)."""
    size = last - first
    unphased = []
    first_seq, second_seq = [""] * size, [""] * size
    idx_first, idx_second = [0] * size, [0] * size
    count = 0
    for i in range(first, last):  # :110-149
        rec = all_variants[sorted_index[i]]
        if rec[0].allele == -1:
            unphased.append(count)
        else:
            k = i - first
            if len(rec) == 2:
                first_seq[k], idx_first[k] = rec[0].alt, 0
                second_seq[k], idx_second[k] = rec[1].alt, 1
            elif rec[0].allele == 0:
                first_seq[k], idx_first[k] = rec[0].alt, 0
                second_seq[k], idx_second[k] = rec[0].ref, -1
            elif rec[0].allele == 1:
                first_seq[k], idx_first[k] = rec[0].ref, -1
                second_seq[k], idx_second[k] = rec[0].alt, 0
            else:
                first_seq[k], idx_first[k] = rec[0].alt, 0
                second_seq[k] = rec[0].alt
                idx_first[k] = 0  # :145 assigns indexVariantsFirst again; indexVariantsSecond stays 0
        count += 1
    combos, ids = [], []
    if unphased:  # :155-214
        stack = [-1]
        while stack:
            stack[-1] += 1
            if stack[-1] >= 2:
                stack.pop()
            elif len(stack) < len(unphased):
                stack.append(-1)
            else:
                for i in range(len(stack)):
                    u = unphased[i]
                    rec = all_variants[sorted_index[first + u]]
                    if len(rec) == 2:
                        first_seq[u] = second_seq[u] = rec[stack[i]].alt
                        idx_first[u] = idx_second[u] = stack[i]
                    elif stack[i] == 0:
                        first_seq[u] = second_seq[u] = rec[0].ref
                        idx_first[u] = idx_second[u] = -1
                    else:
                        first_seq[u] = second_seq[u] = rec[0].alt
                        idx_first[u] = idx_second[u] = 0
                combos.append(list(first_seq))
                ids.append(get_fasta_id(all_variants, sorted_index, first, index_center, idx_first, chr_name))
                if idx_first != idx_second:
                    combos.append(list(second_seq))
                    ids.append(get_fasta_id(all_variants, sorted_index, first, index_center, idx_second, chr_name))
    else:  # :215-228
        combos.append(list(first_seq))
        ids.append(get_fasta_id(all_variants, sorted_index, first, index_center, idx_first, chr_name))
        if idx_first != idx_second:
            combos.append(list(second_seq))
            ids.append(get_fasta_id(all_variants, sorted_index, first, index_center, idx_second, chr_name))
    return combos, ids


def extract(genome, chr_name, start, end):
    """write_fasta.h:245-271: clamped FAI region read.  genome: {first word of the FASTA id: sequence}."""
    seq = genome[chr_name]
    start = min(start, len(seq))
    end = min(end, len(seq))
    if start > end:
        end = start
    return dna5(seq[start:end])


def generate_variant_sequences(genome, all_variants, sorted_index, chr_name, rng, index_center):
    """write_fasta.h:303-399.  Returns (sequences, ids)."""
    i1, i2 = rng
    center = all_variants[index_center][0]
    start_variant = center.start > all_variants[sorted_index[i1]][0].pos  # :315-318
    end_variant = center.end == all_variants[sorted_index[i2 - 1]][0].pos  # :320-323
    if start_variant and end_variant:
        n, ref_start, r_start, r_end = 2 * (i2 - i1) - 1, 1, i1 + 1, i2
    elif start_variant:
        n, ref_start, r_start, r_end = 2 * (i2 - i1), 1, i1 + 1, i2 + 1
    elif end_variant:
        n, ref_start, r_start, r_end = 2 * (i2 - i1), 0, i1, i2
    else:
        n, ref_start, r_start, r_end = 2 * (i2 - i1) + 1, 0, i1, i2 + 1
    base = [""] * n
    j = ref_start
    for i in range(r_start, r_end):  # :366-384
        if j == 0:
            b, e = center.start, all_variants[sorted_index[i]][0].pos
        elif i == i2:
            prev = all_variants[sorted_index[i - 1]][0]
            b, e = prev.pos + len(prev.ref), center.end
        else:
            prev = all_variants[sorted_index[i - 1]][0]
            b, e = prev.pos + len(prev.ref), all_variants[sorted_index[i]][0].pos
        base[j] = extract(genome, chr_name, b, e)
        j += 2
    combos, ids = all_combinations(all_variants, sorted_index, i1, i2, index_center, chr_name)
    seqs = []
    for combo in combos:  # :391-398
        k = 1 - ref_start
        for alt in combo:
            base[k] = alt
            k += 2
        seqs.append("".join(base))
    return seqs, ids


def vcf_loader(vcf_text, genome, sample_index=0, seq_length=23):
    """vcf_loader.cpp:40-68: returns the SNP-genome FASTA records [(id, sequence)] in file order."""
    all_variants, chr_table = process_vcf_text(vcf_text, sample_index)
    regions, centers, sorted_index = get_variant_overlap_ranges(all_variants, len(chr_table), seq_length)
    out = []
    for c in range(len(regions)):  # write_fasta.h:453-463
        for j in range(len(regions[c])):
            seqs, ids = generate_variant_sequences(genome, all_variants, sorted_index[c], chr_table[c], regions[c][j],
                                                   centers[c][j])
            out += list(zip(ids, seqs))
    return out


def format_fasta(records, width=70):
    """SeqAn SeqFileOut default: '>' id, sequence wrapped at 70 characters."""
    lines = []
    for rid, seq in records:
        lines.append(">" + rid)
        for i in range(0, len(seq), width):
            lines.append(seq[i:i + width])
        if not seq:
            lines.append("")
    return "\n".join(lines) + ("\n" if lines else "")
