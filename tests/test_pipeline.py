"""End-to-end drop-in pipeline on the GPU box: bidir_index -> bidir_mapping (reference and SNP
genome) -> vcf_loader -> bam_merger_ref_only / bam_merger, every stage's text output compared with
the oracle's restatement of the reference stage."""
import os
import subprocess

import numpy as np
import pytest

from helpers import mutate, random_seq, revcomp
from oracle import merge_oracle as mo
from oracle import variants_oracle as vo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("VSC_TEST_BIN") or os.path.join(ROOT, "varscot_amd", "bin")  # (tools/sanitize_cpu.sh: sanitizer builds)


def run(tool, *args):
    r = subprocess.run([os.path.join(BIN, tool)] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (tool, r.stdout, r.stderr)
    return r


def write_fasta(path, records):
    with open(path, "w") as f:
        for name, seq in records:
            f.write(">%s\n" % name)
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")


def read_fasta(path):
    recs, name = [], None
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:]
            recs.append([name, ""])
        elif line:
            recs[-1][1] += line
    return [(a, b) for a, b in recs]


@pytest.fixture(scope="module")
def scenario(tmp_path_factory):
    return build_scenario(tmp_path_factory.mktemp("pipeline"))


def build_scenario(d, seed=20240):
    """A small genome with on-targets, planted off-targets and variants next to them, written into directory d
    (also what tests/test_mergers.py feeds the host tools with, from the oracle's SAM text)."""
    rng = np.random.default_rng(seed)
    contigs = {"chr1": random_seq(rng, 30000), "chr2": random_seq(rng, 18000), "chrM": random_seq(rng, 3000)}
    # on-targets: 23-mers ending in GG taken from the genome (forward) or placed as reverse complement
    targets = []
    for i, (c, p, strand) in enumerate([("chr1", 5000, "+"), ("chr1", 12000, "-"), ("chr2", 3000, "+"),
                                        ("chr2", 9000, "-"), ("chr1", 20000, "+")]):
        g = random_seq(rng, 21) + "GG"
        site = g if strand == "+" else revcomp(g)
        s = contigs[c]
        contigs[c] = s[:p] + site + s[p + 23:]
        targets.append(("site%d" % i, c, p, strand, g))
    # off-targets: mutated copies, some of them next to variants
    planted = []
    for k in range(60):
        name, c0, p0, st0, g = targets[k % len(targets)]
        c = ["chr1", "chr2", "chrM"][k % 3]
        p = 200 + 400 * k % (len(contigs[c]) - 300)
        site = mutate(rng, g, int(rng.integers(0, 5)), 0, 20)
        if k % 2:
            site = revcomp(site)
        s = contigs[c]
        if any(c == tc and abs(p - tp) < 60 for _, tc, tp, _, _ in targets):
            continue
        contigs[c] = s[:p] + site + s[p + 23:]
        planted.append((c, p))
    # variants: SNPs and small indels inside / next to planted sites
    vcf = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0"]
    gts = ["0|1", "1|0", "1|1", "0/1"]
    used = set()
    for k, (c, p) in enumerate(planted[:40]):
        q = p + int(rng.integers(-5, 30))
        if q < 30 or q > len(contigs[c]) - 40 or any(abs(q - u) < 3 and uc == c for uc, u in used):
            continue
        used.add((c, q))
        ref = contigs[c][q]
        kind = k % 5
        if kind == 3:
            r, a = contigs[c][q:q + 3], ref              # deletion of 2
        elif kind == 4:
            r, a = ref, ref + random_seq(rng, 2)         # insertion of 2
        else:
            r, a = ref, [x for x in "ACGT" if x != ref][k % 3]
        vcf.append("%s\t%d\t.\t%s\t%s\t.\tPASS\t.\tGT\t%s" % (c, q + 1, r, a, gts[k % 4]))
    records = [(n, s) for n, s in contigs.items()]
    write_fasta(d / "genome.fa", records)
    (d / "in.vcf").write_text("\n".join(vcf) + "\n")
    bed = "".join("%s\t%d\t%d\t%s\t7\t%s\n" % (c, p, p + 23, name, strand) for name, c, p, strand, g in targets)
    (d / "targets.bed").write_text(bed)
    write_fasta(d / "targets.fa", [(name, g) for name, c, p, strand, g in targets])
    tus = "ID   Sequence   Score   Dir\n" + "".join("%s %s %.6f +\n" % (name, "A" * 30, 0.5 + 0.25 * i)
                                                    for i, (name, *_r) in enumerate(targets))
    (d / "activity.txt").write_text(tus)
    return d, records, bed, tus, targets


def test_reference_only_pipeline(scenario):
    d, records, bed, tus, targets = scenario
    run("bidir_index", "-G", d / "genome.fa", "-I", d / "ref_idx")
    run("bidir_mapping", "-G", d / "genome.fa", "-I", d / "ref_idx", "-R", d / "targets.fa", "-M", 5, "-O", d / "ref.sam")
    sam = (d / "ref.sam").read_text()
    assert len(sam.splitlines()) > 30
    for mit in (0, 1):
        run("bam_merger_ref_only", d / "out.txt", d / "feat.txt", d / "ref.sam", d / "targets.bed", d / "genome.fa",
            d / "activity.txt", 5, 23, mit)
        want_tsv, want_fm = mo.process_ref_only(sam, bed, records, tus, bool(mit))
        assert (d / "out.txt").read_text() == want_tsv
        if mit:
            assert (d / "feat.txt").read_text() == want_fm
    # the on-target loci themselves are not reported
    tsv = (d / "out.txt").read_text().splitlines()[1:]
    for name, c, p, strand, g in targets:
        assert not any(l.split("\t")[0] == c and int(l.split("\t")[1]) == p and l.split("\t")[3].startswith(name + "_")
                       and l.split("\t")[7] == "0" for l in tsv)


def test_variant_aware_pipeline(scenario):
    d, records, bed, tus, targets = scenario
    run("vcf_loader", d / "in.vcf", d / "snp.fa", d / "genome.fa", 0, 23, 4)
    genome = dict(records)
    assert (d / "snp.fa").read_text() == vo.format_fasta(vo.vcf_loader((d / "in.vcf").read_text(), genome, 0, 23))
    snp_records = read_fasta(d / "snp.fa")
    assert len(snp_records) > 20
    run("bidir_index", "-G", d / "genome.fa", "-I", d / "ref_idx")
    run("bidir_mapping", "-G", d / "genome.fa", "-I", d / "ref_idx", "-R", d / "targets.fa", "-M", 5, "-O", d / "ref.sam")
    run("bidir_index", "-G", d / "snp.fa", "-I", d / "snp_idx")
    run("bidir_mapping", "-G", d / "snp.fa", "-I", d / "snp_idx", "-R", d / "targets.fa", "-M", 5, "-O", d / "snp.sam")
    ref_sam, snp_sam = (d / "ref.sam").read_text(), (d / "snp.sam").read_text()
    assert len(snp_sam.splitlines()) > 10
    for mit in (0, 1):
        run("bam_merger", d / "merged.txt", d / "mfeat.txt", d / "ref.sam", d / "snp.sam", d / "targets.bed", d / "genome.fa",
            d / "snp.fa", d / "activity.txt", 5, 23, 4, mit)
        want_tsv, want_fm = mo.merge_results(ref_sam, snp_sam, bed, records, snp_records, tus, 23, bool(mit))
        assert (d / "merged.txt").read_text() == want_tsv
        if mit:
            assert (d / "mfeat.txt").read_text() == want_fm
    rows = (d / "merged.txt").read_text().splitlines()[1:]
    assert any(r.split("\t")[-1].startswith("VAR_") for r in rows) and any(r.split("\t")[-1] == "REF" for r in rows)


def test_mergers_take_reference_bases_from_the_packed_genomes(scenario):
    """The mergers with the packed genomes of `bidir_index` in place of the two FASTA files they are given (reference:
    VARSCOT_PACKED_GENOME, SNP genome: VARSCOT_PACKED_SNP_GENOME - what the driver exports): the hit sequences, the on-target
    sequences and the windows of the shadow filter come from <prefix>.vsc; result and feature matrix are the restatement's."""
    d, records, bed, tus, targets = scenario
    run("vcf_loader", d / "in.vcf", d / "snp.fa", d / "genome.fa", 0, 23, 4)
    run("bidir_index", "-G", d / "genome.fa", "-I", d / "ref_idx")
    run("bidir_index", "-G", d / "snp.fa", "-I", d / "snp_idx")
    run("bidir_mapping", "-G", d / "genome.fa", "-I", d / "ref_idx", "-R", d / "targets.fa", "-M", 5, "-O", d / "ref.sam")
    run("bidir_mapping", "-G", d / "snp.fa", "-I", d / "snp_idx", "-R", d / "targets.fa", "-M", 5, "-O", d / "snp.sam")
    env = dict(os.environ, VARSCOT_TRACE="1", VARSCOT_PACKED_GENOME=str(d / "ref_idx"), VARSCOT_PACKED_SNP_GENOME=str(d / "snp_idx"))
    ref_sam, snp_sam = (d / "ref.sam").read_text(), (d / "snp.sam").read_text()
    snp_records = read_fasta(d / "snp.fa")
    r = subprocess.run([os.path.join(BIN, "bam_merger"), str(d / "pm.txt"), str(d / "pmf.txt"), str(d / "ref.sam"), str(d / "snp.sam"),
                        str(d / "targets.bed"), str(d / "genome.fa"), str(d / "snp.fa"), str(d / "activity.txt"), "5", "23", "4", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stderr.count("from the packed genome") == 2 and "FASTA text" not in r.stderr
    want_tsv, want_fm = mo.merge_results(ref_sam, snp_sam, bed, records, snp_records, tus, 23, True)
    assert (d / "pm.txt").read_text() == want_tsv and (d / "pmf.txt").read_text() == want_fm
    r = subprocess.run([os.path.join(BIN, "bam_merger_ref_only"), str(d / "pr.txt"), str(d / "prf.txt"), str(d / "ref.sam"),
                        str(d / "targets.bed"), str(d / "genome.fa"), str(d / "activity.txt"), "5", "23", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stderr.count("from the packed genome") == 1, r.stdout + r.stderr
    assert (d / "pr.txt").read_text() == mo.process_ref_only(ref_sam, bed, records, tus, False)[0]


@pytest.mark.parametrize("evaluation", ["mit", "prob", "class"])
def test_driver_in_one_process_equals_the_staged_tools(scenario, tmp_path, evaluation):
    """The driver's default route (varscot_pipeline: on-targets from the packed genome, search, windows straight from the planes,
    merge, scores, forest and the final sort in ONE process - no SAM text, no SNP-genome FASTA, no feature matrix parsed back)
    against the staged route (VARSCOT_STAGED=1: fasta_writer | bidir_mapping | vcf_loader | bidir_index | bidir_mapping |
    bam_merger[_ref_only] | classification_pipeline | sort, as VARSCOT:260-357 chains them): result and feature matrix byte for
    byte, without a VCF, with one sample column and with two."""
    d, records, bed, tus, targets = scenario
    driver = os.path.join(ROOT, "varscot_amd", "driver", "VARSCOT")
    lines, k = [], 0
    other = ["0|0", "1|1", "0|1", "1|0", "./.", "1/1"]
    for l in (d / "in.vcf").read_text().splitlines():
        if l.startswith("##"):
            lines.append(l)
        elif l.startswith("#CHROM"):
            lines.append(l + "\tS1")
        else:
            lines.append(l + "\t" + other[k % len(other)])
            k += 1
    vcf2 = tmp_path / "two.vcf"
    vcf2.write_text("\n".join(lines) + "\n")
    for case, extra in (("ref", []), ("one", ["-f", str(d / "in.vcf"), "-s", "0"]), ("two", ["-f", str(vcf2), "-s", "all"]), ("pam", ["-p", "AG"])):
        got = {}
        for route in ("inproc", "staged") + (("shards",) if case == "one" else ()):
            out = tmp_path / ("%s_%s.txt" % (case, route))
            cmd = ["bash", driver, "-b", str(d / "targets.bed"), "-o", str(out), "-g", str(d / "genome.fa"), "-i", str(tmp_path / "idx"),
                   "-m", "5", "-t", "2", "-T", str(tmp_path / ("tmp_" + route)), "-a", str(d / "activity.txt"), "-e", evaluation] + extra
            env = dict(os.environ)
            if route == "staged":
                env["VARSCOT_STAGED"] = "1"
            if route == "shards":  # the reference search over three contexts on the one GPU (vsc_multi_search), same files
                env["VARSCOT_DEVICES"] = "0,0,0"
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stdout + r.stderr
            stems = ["%s_sample%d" % (str(out)[:-4], k) for k in (0, 1)] if case == "two" else [str(out)[:-4]]
            files = {}
            for stem in stems:
                files[os.path.basename(stem).replace(route, "")] = open(stem + ".txt", "rb").read()
                if evaluation != "mit":
                    files[os.path.basename(stem).replace(route, "") + "_fm"] = open(stem + "_feature_matrix.txt", "rb").read()
                else:
                    assert not os.path.exists(stem + "_feature_matrix.txt")
            got[route] = files
        assert got["inproc"] == got["staged"], case
        assert got.get("shards", got["inproc"]) == got["inproc"], case
        first = next(iter(got["inproc"].values())).decode().splitlines()
        assert len(first) > 20 and first[0].split("\t")[3] == ("Targetsite" if evaluation == "mit" else "Name")
        if case != "ref" and case != "pam":
            assert any(l.endswith("REF") for l in first[1:]) and any("VAR_" in l.split("\t")[-1] for l in first[1:])


def test_driver_end_to_end(scenario, tmp_path):
    """The VARSCOT driver (same flags as the reference's) over the drop-in tools, with and without a
    VCF; its TSV must equal the merger's rows sorted on the name column."""
    d, records, bed, tus, targets = scenario
    driver = os.path.join(ROOT, "varscot_amd", "driver", "VARSCOT")
    for vcf in (None, d / "in.vcf"):
        out = tmp_path / ("res_%s.txt" % ("vcf" if vcf else "ref"))
        cmd = ["bash", driver, "-b", str(d / "targets.bed"), "-o", str(out), "-g", str(d / "genome.fa"), "-i",
               str(tmp_path / "idx"), "-m", "5", "-t", "2", "-T", str(tmp_path / "tmp"), "-a", str(d / "activity.txt")]
        if vcf:
            cmd += ["-f", str(vcf), "-s", "0"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = out.read_text().splitlines()
        assert lines[0].startswith("#Chr\tStart\tEnd\tTargetsite\tScore")
        names = [l.split("\t")[3] for l in lines[1:]]
        assert names == sorted(names) and len(names) > 20
        assert not (tmp_path / "tmp").exists()  # temp dir removed unless -v
    # -e prob: the random forest replaces the MIT score (classification_pipeline on the GPU)
    out = tmp_path / "res_prob.txt"
    r = subprocess.run(["bash", driver, "-b", str(d / "targets.bed"), "-o", str(out), "-g", str(d / "genome.fa"), "-i",
                        str(tmp_path / "idx"), "-m", "5", "-T", str(tmp_path / "tmp"), "-a", str(d / "activity.txt"), "-e", "prob"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    scores = [float(l.split("\t")[4]) for l in out.read_text().splitlines()[1:]]
    assert len(scores) > 20 and all(0.0 <= s <= 1.0 for s in scores)
    # the feature matrix stays next to the result, as the reference leaves it (VARSCOT:334-342)
    fm = (tmp_path / "res_prob_feature_matrix.txt").read_text().splitlines()
    assert len(fm) == len(scores) + 1 and fm[0].split("\t")[0] == "totalMismatches"
    r = subprocess.run(["bash", driver, "-b", "x.bed", "-o", "o.txt", "-g", "g.fa", "-i", "i", "-m", "9", "-T", str(tmp_path / "t2")],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "between 0 and 8" in r.stdout
    r = subprocess.run(["bash", driver, "-o", "o.txt", "-T", str(tmp_path / "t3")], capture_output=True, text=True)
    assert r.returncode == 2 and "No on-target file path" in r.stdout


def test_driver_several_samples_in_one_run(scenario, tmp_path):
    """-s 0,1 / -s all: what the reference's parallel.py does with one driver process per VCF sample column
    (parallel.py:49-63) in one run that shares the reference search.  Every sample's result and feature matrix must be
    byte-identical to a single-sample run of that column."""
    d, records, bed, tus, targets = scenario
    driver = os.path.join(ROOT, "varscot_amd", "driver", "VARSCOT")
    # a second sample column with other genotypes (some sites absent, some homozygous)
    other = ["0|0", "1|1", "0|1", "1|0", "./.", "1/1"]
    lines, k = [], 0
    for l in (d / "in.vcf").read_text().splitlines():
        if l.startswith("##"):
            lines.append(l)
        elif l.startswith("#CHROM"):
            lines.append(l + "\tS1")
        else:
            lines.append(l + "\t" + other[k % len(other)])
            k += 1
    vcf2 = tmp_path / "two.vcf"
    vcf2.write_text("\n".join(lines) + "\n")
    base = ["bash", driver, "-b", str(d / "targets.bed"), "-g", str(d / "genome.fa"), "-i", str(tmp_path / "idx"), "-m", "5",
            "-t", "2", "-a", str(d / "activity.txt"), "-f", str(vcf2), "-e", "prob"]
    single = {}
    for s in (0, 1):
        out = tmp_path / ("one_%d.txt" % s)
        r = subprocess.run(base + ["-o", str(out), "-T", str(tmp_path / "tmp1"), "-s", str(s)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        single[s] = (out.read_bytes(), (tmp_path / ("one_%d_feature_matrix.txt" % s)).read_bytes())
    assert single[0][0] != single[1][0]  # the columns differ, so do the results
    for spec in ("0,1", "all"):
        out = tmp_path / ("multi_%s.txt" % spec.replace(",", ""))
        r = subprocess.run(base + ["-o", str(out), "-T", str(tmp_path / "tmp2"), "-s", spec], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("Searching for off-targets") == 1
        stem = str(out)[:-4]
        for s in (0, 1):
            assert open("%s_sample%d.txt" % (stem, s), "rb").read() == single[s][0]
            assert open("%s_sample%d_feature_matrix.txt" % (stem, s), "rb").read() == single[s][1]
    r = subprocess.run(base + ["-o", str(tmp_path / "bad.txt"), "-T", str(tmp_path / "tmp3"), "-s", "0,x"], capture_output=True, text=True)
    assert r.returncode == 2 and "Sample must be" in r.stdout
