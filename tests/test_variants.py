"""Alt-allele expansion (row R8): the C++ `vcf_loader` drop-in against the pure-Python restatement of
variant_processing/{process_vcf,overlap_sequences,write_fasta}.h on seeded synthetic VCFs."""
import os
import subprocess

import numpy as np
import pytest

from helpers import random_seq
from oracle import variants_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(os.environ.get("VSC_TEST_BIN") or os.path.join(ROOT, "varscot_amd", "bin"), "vcf_loader")


def synth_vcf(seed, genome, n_records, n_samples=2, header_contigs=None, indel_rate=0.3, cluster=True):
    rng = np.random.default_rng(seed)
    names = list(genome)
    lines = ["##fileformat=VCFv4.2"]
    for c in header_contigs or []:
        lines.append("##contig=<ID=%s,length=%d>" % (c, len(genome[c])))
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("S%d" % i for i in range(n_samples)))
    gts = ["0|1", "1|0", "1|1", "0/1", "1/1", "0|0", "./.", "1|2", "2|1", "1/2", "0|2", "1", ".|1", "1|.", "2/2"]
    recs = []
    for _ in range(n_records):
        c = names[int(rng.integers(0, len(names)))]
        L = len(genome[c])
        if cluster and recs and rng.random() < 0.5 and recs[-1][0] == c:
            pos = min(L - 12, recs[-1][1] + int(rng.integers(1, 30)))  # chains of nearby variants
        else:
            pos = int(rng.integers(1, L - 12))
        kind = rng.random()
        ref = genome[c][pos - 1]
        if ref == "N":
            continue
        def other(b):
            return rng.choice([x for x in "ACGT" if x != b])
        if kind < indel_rate / 2:      # deletion
            ln = int(rng.integers(1, 9))
            r, a = genome[c][pos - 1:pos + ln], ref
        elif kind < indel_rate:        # insertion
            r, a = ref, ref + random_seq(rng, int(rng.integers(1, 7)))
        else:
            r, a = ref, other(ref)
        alts = [a]
        if rng.random() < 0.25:        # second alt allele
            alts.append(r[0] + random_seq(rng, int(rng.integers(1, 4))) if rng.random() < 0.5 else other(r[0]))
        if rng.random() < 0.03:
            alts[0] = "."
        fmt = "GT:DP" if rng.random() < 0.7 else "DP:GT"
        cols = []
        for _s in range(n_samples):
            gt = gts[int(rng.integers(0, len(gts)))]
            if len(alts) == 1:
                gt = gt.replace("2", "1")
            cols.append(gt + ":7" if fmt == "GT:DP" else "7:" + gt)
        recs.append((c, pos, "%s\t%d\t.\t%s\t%s\t.\tPASS\t.\t%s\t%s" % (c, pos, r, ",".join(alts), fmt, "\t".join(cols))))
    order = rng.permutation(len(recs))  # the sweep sorts by position itself
    lines += [recs[i][2] for i in order]
    return "\n".join(lines) + "\n"


def run_tool(tmp_path, vcf_text, genome, sample):
    (tmp_path / "in.vcf").write_text(vcf_text)
    with open(tmp_path / "genome.fa", "w") as f:
        for name, seq in genome.items():
            f.write(">%s some description\n" % name)
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")
    r = subprocess.run([BIN, str(tmp_path / "in.vcf"), str(tmp_path / "snp.fa"), str(tmp_path / "genome.fa"), str(sample),
                        "23", "2"], capture_output=True, text=True, timeout=300)
    return r, (tmp_path / "snp.fa").read_text() if (tmp_path / "snp.fa").exists() else None


@pytest.mark.parametrize("seed,sample,indel_rate", [(1, 0, 0.0), (2, 1, 0.0), (3, 0, 0.3), (4, 1, 0.5), (5, 0, 0.8), (6, 0, 0.3)])
def test_vcf_loader_matches_restatement(tmp_path, seed, sample, indel_rate):
    rng = np.random.default_rng(1000 + seed)
    genome = {"chr1": random_seq(rng, 4000), "chr2": random_seq(rng, 2500), "chrUn_x": random_seq(rng, 600)}
    genome["chr2"] = genome["chr2"][:700] + "N" * 40 + genome["chr2"][740:]
    vcf = synth_vcf(seed, genome, 260, header_contigs=["chr2", "chr1"] if seed % 2 else None, indel_rate=indel_rate)
    r, got = run_tool(tmp_path, vcf, genome, sample)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.splitlines() == ["Process records", "Compute overlap sequences", "Write fasta"]
    want = vo.format_fasta(vo.vcf_loader(vcf, genome, sample, 23))
    assert got.count(">") > 100
    assert got == want


def test_isolated_snp_window_and_ids(tmp_path):
    """An isolated SNP gives the 45-base window [pos-22, pos+23) in a REF and an ALT version
    (SURVEY.md 8.3); ids are chr_start_REF / chr_start_ALT_pos_ref_alt with 0-based positions."""
    rng = np.random.default_rng(9)
    genome = {"chrA": random_seq(rng, 300)}
    ref = genome["chrA"][99]
    alt = "A" if ref != "A" else "C"
    vcf = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\nchrA\t100\t.\t%s\t%s\t.\t.\t.\tGT\t0|1\n" % (ref, alt)
    r, got = run_tool(tmp_path, vcf, genome, 0)
    assert r.returncode == 0
    w = genome["chrA"][77:122]
    assert got == ">chrA_77_REF\n%s\n>chrA_77_ALT_99_%s_%s\n%s\n" % (w, ref, alt, w[:22] + alt + w[23:])


def test_variant_near_contig_start_wraps_like_the_reference(tmp_path):
    rng = np.random.default_rng(10)
    genome = {"c": random_seq(rng, 200)}
    ref = genome["c"][4]
    alt = "G" if ref != "G" else "T"
    vcf = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\nc\t5\t.\t%s\t%s\t.\t.\t.\tGT\t1|1\n" % (ref, alt)
    r, got = run_tool(tmp_path, vcf, genome, 0)
    want = vo.format_fasta(vo.vcf_loader(vcf, genome, 0, 23))
    assert r.returncode == 0 and got == want
    assert got.startswith(">c_%d_ALT_4_" % ((4 - 23 + 1) % (1 << 32)))  # unsigned start, overlap_sequences.h:158


def test_cli_errors(tmp_path):
    assert subprocess.run([BIN], capture_output=True).returncode == 1
    r = subprocess.run([BIN, "a.vcf", "o.fa", "g.fa", "x", "23", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot cast x into an unsigned" in r.stderr
    r = subprocess.run([BIN, str(tmp_path / "missing.vcf"), str(tmp_path / "o.fa"), str(tmp_path / "g.fa"), "0", "23", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 1


def _fasta_route(tmp_path, vcf, genome, sample):
    """vcf_loader -> FASTA -> packed planes: the route the windows-from-planes builder must reproduce."""
    import varscot_amd as va
    r, text = run_tool(tmp_path, vcf, genome, sample)
    assert r.returncode == 0, r.stdout + r.stderr
    names, seqs = [], []
    for block in text.split(">")[1:]:
        head, _, body = block.partition("\n")
        names.append(head)
        seqs.append(body.replace("\n", ""))
    return va.PackedGenome.from_sequences(seqs, names)


@pytest.mark.parametrize("seed,sample,indel_rate,threads", [(1, 0, 0.0, 1), (2, 1, 0.0, 3), (3, 0, 0.3, 2), (4, 1, 0.5, 8),
                                                           (5, 0, 0.8, 5), (6, 0, 0.3, 0)])
def test_windows_from_planes_equal_the_fasta_route(tmp_path, seed, sample, indel_rate, threads):
    """vsc_windows_build (reference segments copied bit-wise from the packed reference planes, parsing and
    assembly on several threads) gives the planes, contig table and ids of vcf_loader + bidir_index, byte for
    byte: SNPs, indels, multi-allelic and unphased records, N runs, variants near contig ends."""
    import varscot_amd as va
    rng = np.random.default_rng(1000 + seed)
    genome = {"chr1": random_seq(rng, 4000), "chr2": random_seq(rng, 2500), "chrUn_x": random_seq(rng, 600)}
    genome["chr2"] = genome["chr2"][:700] + "N" * 40 + genome["chr2"][740:]
    vcf = synth_vcf(seed, genome, 260, header_contigs=["chr2", "chr1"] if seed % 2 else None, indel_rate=indel_rate)
    want = _fasta_route(tmp_path, vcf, genome, sample)
    ref = va.PackedGenome.from_sequences(list(genome.values()), [n + " some description" for n in genome])
    got = va.variant_windows(ref, tmp_path / "in.vcf", sample=sample, threads=threads)
    assert len(got.contigs) == len(want.contigs) > 100
    assert got.contigs.tobytes() == want.contigs.tobytes()
    assert list(got.names) == want.names
    for a, b in ((got.hi, want.hi), (got.lo, want.lo), (got.nmask, want.nmask)):
        assert a.tobytes() == b.tobytes()


def test_windows_from_planes_many_blocks(tmp_path):
    """More ranges than one work unit takes (4 096), so that the units' bit streams are stitched at arbitrary
    bit offsets by several threads; plus a chromosome the genome does not have (an error, as in the tool)."""
    import varscot_amd as va
    rng = np.random.default_rng(77)
    genome = {"chrA": random_seq(rng, 900_000), "chrB": random_seq(rng, 300_000)}
    vcf = synth_vcf(11, genome, 30_000, n_samples=1, indel_rate=0.1, cluster=False)
    want = _fasta_route(tmp_path, vcf, genome, 0)
    ref = va.PackedGenome.from_sequences(list(genome.values()), list(genome))
    got = va.variant_windows(ref, tmp_path / "in.vcf", sample=0, threads=4)
    assert len(got.contigs) == len(want.contigs) > 20_000
    assert got.contigs.tobytes() == want.contigs.tobytes()
    assert got.hi.tobytes() == want.hi.tobytes() and got.lo.tobytes() == want.lo.tobytes() and got.nmask.tobytes() == want.nmask.tobytes()
    assert got.names[0] == want.names[0] and got.names[len(want.names) - 1] == want.names[-1]
    assert got.names[12345] == want.names[12345]
    small = va.PackedGenome.from_sequences([genome["chrA"]], ["chrA"])
    with pytest.raises(va.VarscotError):
        va.variant_windows(small, tmp_path / "in.vcf", sample=0)


def _unphased_cluster(genome_seq, first_pos, n):
    """n unphased heterozygous SNPs at consecutive positions from first_pos (1-based) on contig c."""
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0"]
    for i in range(n):
        ref = genome_seq[first_pos - 1 + i]
        lines.append("c\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t0/1" % (first_pos + i, ref, "A" if ref != "A" else "C"))
    return "\n".join(lines) + "\n"


def test_unphased_records_of_one_window_are_enumerated_up_to_a_limit(tmp_path):
    """write_fasta.h:155-229 writes 2^n windows for n unphased records in one range.  Seven of them: 128 allele combinations per
    range, both routes equal to the restatement; 25: the reference would hold 2^25 sequences in memory before writing one,
    a streaming port would fill the disk - refused with a message by the tool and by vsc_windows_build."""
    import varscot_amd as va
    rng = np.random.default_rng(4242)
    genome = {"c": random_seq(rng, 400)}
    vcf = _unphased_cluster(genome["c"], 150, 7)
    r, got = run_tool(tmp_path, vcf, genome, 0)
    assert r.returncode == 0, r.stdout + r.stderr
    assert got == vo.format_fasta(vo.vcf_loader(vcf, genome, 0, 23)) and got.count(">") >= 128
    want = _fasta_route(tmp_path, vcf, genome, 0)
    ref = va.PackedGenome.from_sequences(list(genome.values()), [n + " some description" for n in genome])
    win = va.variant_windows(ref, tmp_path / "in.vcf", sample=0, threads=3)
    assert list(win.names) == want.names and win.hi.tobytes() == want.hi.tobytes() and win.lo.tobytes() == want.lo.tobytes()
    vcf = _unphased_cluster(genome["c"], 150, 25)
    r, got = run_tool(tmp_path, vcf, genome, 0)
    assert r.returncode == 1 and "25 unphased variants within one window" in r.stdout, r.stdout + r.stderr
    with pytest.raises(Exception, match="25 unphased variants within one window"):
        va.variant_windows(ref, tmp_path / "in.vcf", sample=0, threads=3)
