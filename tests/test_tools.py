"""Command-line drop-ins (varscot_amd/bin): flags, messages and exit codes of the reference's
read_mapping tools; on the GPU box the SAM text is compared byte for byte with the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from helpers import make_genome, random_guides

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.environ.get("VSC_TEST_BIN") or os.path.join(ROOT, "varscot_amd", "bin")  # (tools/sanitize_cpu.sh: sanitizer builds)


def run(tool, *args):
    return subprocess.run([os.path.join(BIN, tool)] + list(args), capture_output=True, text=True, timeout=600)


def write_fasta(path, names, seqs, width=60):
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n" % n)
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")


@pytest.fixture()
def workdir(tmp_path):
    rng = np.random.default_rng(314)
    guides = random_guides(rng, 7)
    contigs = make_genome(314, [12000, 5000, 30, 2500], guides, 6, n_plant=150, n_runs=3)
    names = ["chr1 primary assembly", "chr2", "tiny", "chrUn_gl000220"]
    gnames = ["site%d" % i for i in range(len(guides))]
    write_fasta(tmp_path / "genome.fa", names, contigs)
    write_fasta(tmp_path / "reads.fa", gnames, guides)
    return tmp_path, names, contigs, gnames, guides


def test_bidir_index_cli(workdir):
    d, names, contigs, _, _ = workdir
    r = run("bidir_index", "-G", str(d / "genome.fa"), "-I", str(d / "idx"))
    assert r.returncode == 0
    assert r.stdout.splitlines() == ["Number of sequences: 4", "Index created successfully"]
    assert (d / "idx.vsc").exists()
    assert run("bidir_index", "-G", str(d / "genome.fa")).returncode == 1          # missing -I
    assert run("bidir_index", "-G", str(d / "genome.txt"), "-I", "x").returncode == 1  # extension check
    assert run("bidir_index", "--help").returncode == 0


def test_tools_take_reference_bases_from_the_packed_genome(workdir):
    """`fasta_writer` and `vcf_loader` read their reference bases from <prefix>.vsc (a few mapped pages
    per region, the part the .fai index plays in the reference: extract_fasta_ontargets.h:33-76, write_fasta.h:245-271) when it
    was packed from the FASTA they are given - found through VARSCOT_PACKED_GENOME, as <fasta>.vsc or as <fasta minus its
    extension>.vsc - and from the FASTA text otherwise (no packed genome, or one whose FASTA has changed since).  Same output
    files every way: lower-case and IUPAC letters, flanks that run over a contig end, '-' records."""
    import shutil
    d, names, contigs, gnames, guides = workdir
    # a genome with soft-masked and ambiguous stretches, ids with descriptions
    seqs = [contigs[0][:300] + contigs[0][300:600].lower() + "RYKM" + contigs[0][604:], contigs[1], contigs[2], contigs[3]]
    write_fasta(d / "g.fa", names, seqs)
    bed = ("chr1\t500\t523\tt0\t0\t+\nchr1\t2\t25\tt1\t0\t-\nchr2\t4977\t5000\tt2\t0\t+\ntiny\t3\t26\tt3\t0\t-\n"
           "chrUn_gl000220\t590\t613\tt4\t0\t+\nchr1\t590\t613\tt5\t0\t-\n")
    (d / "t.bed").write_text(bed)
    vcf = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\n"
    for c, name in ((0, "chr1"), (1, "chr2"), (3, "chrUn_gl000220")):
        for pos in (7, 310, 598, 603, 1200, len(seqs[c]) - 5):
            ref = seqs[c][pos - 1].upper()
            if ref in "ACGT":
                vcf += "%s\t%d\t.\t%s\t%s\t.\t.\t.\tGT\t0|1\n" % (name, pos, ref, "A" if ref != "A" else "C")
    (d / "v.vcf").write_text(vcf)

    def outputs(env, tag):
        e = dict(os.environ, VARSCOT_TRACE="1", **env)
        r1 = subprocess.run([os.path.join(BIN, "fasta_writer"), str(d / ("o1_%s.fa" % tag)), str(d / ("o2_%s.fa" % tag)), str(d / "t.bed"),
                             str(d / "g.fa")], capture_output=True, text=True, env=e)
        r2 = subprocess.run([os.path.join(BIN, "vcf_loader"), str(d / "v.vcf"), str(d / ("snp_%s.fa" % tag)), str(d / "g.fa"), "0", "23", "2"],
                            capture_output=True, text=True, env=e)
        assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr + r1.stdout + r2.stdout
        return [(d / (n % tag)).read_bytes() for n in ("o1_%s.fa", "o2_%s.fa", "snp_%s.fa")], r1.stderr + r2.stderr

    text, trace = outputs({}, "text")
    assert trace.count("from the FASTA text") == 2 and len(text[2]) > 500
    assert run("bidir_index", "-G", str(d / "g.fa"), "-I", str(d / "pk")).returncode == 0
    packed, trace = outputs({"VARSCOT_PACKED_GENOME": str(d / "pk")}, "env")
    assert trace.count("from the packed genome") == 2 and packed == text
    shutil.copy(d / "pk.vsc", d / "g.fa.vsc")          # sidecar next to the FASTA
    side, trace = outputs({}, "side")
    assert trace.count("from the packed genome") == 2 and side == text
    os.remove(d / "g.fa.vsc")
    shutil.copy(d / "pk.vsc", d / "g.vsc")             # the FASTA's name with .vsc in place of its extension
    side, trace = outputs({}, "stem")
    assert trace.count("from the packed genome") == 2 and side == text
    os.remove(d / "g.vsc")
    # a packed genome whose contig table points outside its planes (a damaged file with a valid stamp) is not trusted:
    # the tools fall back to the FASTA text instead of reading past the mapping
    raw = bytearray((d / "pk.vsc").read_bytes())
    assert raw[:8] == b"VSCIDX02"
    raw[40 + 24 + 8:40 + 24 + 12] = (0xFFFFFFF0).to_bytes(4, "little")   # the second contig's length
    (d / "g.fa.vsc").write_bytes(bytes(raw))
    bad, trace = outputs({}, "bad")
    assert trace.count("from the FASTA text") == 2 and bad == text
    os.remove(d / "g.fa.vsc")
    # the FASTA changes after packing (one base; the size stays): the packed genome is no longer taken
    seqs[0] = seqs[0][:511] + ("A" if seqs[0][511] != "A" else "C") + seqs[0][512:]
    write_fasta(d / "g.fa", names, seqs)
    os.utime(d / "g.fa", ns=(os.stat(d / "g.fa").st_atime_ns, os.stat(d / "g.fa").st_mtime_ns + 5_000_000_000))
    stale, trace = outputs({"VARSCOT_PACKED_GENOME": str(d / "pk")}, "stale")
    assert "not used" in trace and trace.count("from the FASTA text") == 2
    assert stale[0] != text[0] and stale[0] == outputs({}, "text2")[0][0]
    # an index file of the older layout (no source stamp) is still read by bidir_mapping's loader, but stands for no FASTA
    import varscot_amd as va
    pg = va.PackedGenome.from_index_file(str(d / "pk"))
    assert pg.names == names and pg.contig_sequence(1) == contigs[1]


def test_bidir_mapping_cli_errors(workdir):
    d, *_ = workdir
    run("bidir_index", "-G", str(d / "genome.fa"), "-I", str(d / "idx"))
    base = ["-G", str(d / "genome.fa"), "-I", str(d / "idx"), "-R", str(d / "reads.fa"), "-O", str(d / "out.sam")]
    r = run("bidir_mapping", *base, "-M", "9")
    assert r.returncode == 1 and "Maximum number of mismatches must lie between 0 and 8" in r.stderr
    assert run("bidir_mapping", *base).returncode == 1  # -M is required
    assert run("bidir_mapping", "--help").returncode == 0
    r = run("bidir_mapping", *base, "-M", "x")
    assert r.returncode == 1


@pytest.mark.gpu
@pytest.mark.parametrize("md_style,pam", [(0, None), (1, "AG")])
def test_bidir_mapping_sam_equals_oracle(workdir, oracle, md_style, pam):
    d, names, contigs, gnames, guides = workdir
    assert run("bidir_index", "-G", str(d / "genome.fa"), "-I", str(d / "idx")).returncode == 0
    args = ["-G", str(d / "genome.fa"), "-I", str(d / "idx"), "-R", str(d / "reads.fa"), "-M", "6", "-T", "4",
            "-O", str(d / "out.sam"), "--md-style", str(md_style)]
    if pam:
        args += ["-P", pam]
    r = run("bidir_mapping", *args)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["Reads loaded (total: 7).", "Index loaded."]
    got = open(d / "out.sam").read()
    want = oracle.search_sam(contigs, names, guides, gnames, 6, pam, md_style)
    assert len(want.splitlines()) > 50
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0"])
def test_bidir_mapping_over_several_device_contexts(workdir, oracle, devices):
    """-D with a device list: the genome sharded over that many contexts inside the bidir_mapping process
    (vsc_multi: per-shard searches on host threads, one gather to the first context, segment merge) - the SAM
    text is the single-device one, byte for byte.  One GPU here, so the ids repeat; the 12 kb + 5 kb genome leaves
    the later shards of five without words."""
    d, names, contigs, gnames, guides = workdir
    assert run("bidir_index", "-G", str(d / "genome.fa"), "-I", str(d / "idx")).returncode == 0
    args = ["-G", str(d / "genome.fa"), "-I", str(d / "idx"), "-R", str(d / "reads.fa"), "-M", "6", "-O", str(d / "out.sam"),
            "-D", devices]
    r = run("bidir_mapping", *args)
    assert r.returncode == 0, r.stderr
    assert open(d / "out.sam").read() == oracle.search_sam(contigs, names, guides, gnames, 6, None, 0)
    assert run("bidir_mapping", *args[:-1], "0,x").returncode == 1


@pytest.mark.gpu
def test_bidir_mapping_loads_a_saved_seed_index(workdir, oracle):
    """bidir_index -S writes <prefix>.vsi (the seed index built on the device), bidir_mapping loads it instead of
    building: same SAM as the oracle's.  A file of another genome is refused with a message and the search builds
    its own index - the SAM stays right."""
    d, names, contigs, gnames, guides = workdir
    r = run("bidir_index", "-G", str(d / "genome.fa"), "-I", str(d / "idx"), "-S")
    assert r.returncode == 0, r.stderr
    assert (d / "idx.vsi").stat().st_size > 80
    args = ["-G", str(d / "genome.fa"), "-I", str(d / "idx"), "-R", str(d / "reads.fa"), "-M", "6", "-O", str(d / "out.sam")]
    r = run("bidir_mapping", *args)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    want = oracle.search_sam(contigs, names, guides, gnames, 6, None, 0)
    assert open(d / "out.sam").read() == want
    # the index of another genome under this prefix
    other = [c[::-1] for c in contigs]
    write_fasta(d / "other.fa", names, other)
    assert run("bidir_index", "-G", str(d / "other.fa"), "-I", str(d / "other"), "-S").returncode == 0
    os.replace(d / "other.vsi", d / "idx.vsi")
    r = run("bidir_mapping", *args)
    assert r.returncode == 0 and "belongs to another genome" in r.stderr
    assert open(d / "out.sam").read() == want
    # truncated file
    data = (d / "idx.vsi").read_bytes()
    (d / "idx.vsi").write_bytes(data[:40])
    r = run("bidir_mapping", *args)
    assert r.returncode == 0 and "not a seed index file" in r.stderr and open(d / "out.sam").read() == want


@pytest.mark.gpu
def test_bidir_mapping_unwritable_output(workdir):
    d, *_ = workdir
    run("bidir_index", "-G", str(d / "genome.fa"), "-I", str(d / "idx"))
    r = run("bidir_mapping", "-G", str(d / "genome.fa"), "-I", str(d / "idx"), "-R", str(d / "reads.fa"), "-M", "2",
            "-O", str(d / "no_such_dir" / "out.sam"))
    assert r.returncode == 1 and "Could not open output path" in r.stderr


def test_fasta_writer(tmp_path):
    """BED6 -> 23-mers and 30-mers with strand-aware flanks (extract_fasta_ontargets.h:44-53)."""
    from helpers import random_seq, revcomp
    rng = np.random.default_rng(4)
    seq = random_seq(rng, 500)
    write_fasta(tmp_path / "g.fa", ["chrZ extra words"], [seq])
    (tmp_path / "t.bed").write_text("chrZ\t100\t123\tfwd\t7\t+\nchrZ\t200\t223\trev\t7\t-\nchrZ\t2\t25\tedge\t1\t+\n")
    r = run("fasta_writer", str(tmp_path / "a.fa"), str(tmp_path / "b.fa"), str(tmp_path / "t.bed"), str(tmp_path / "g.fa"))
    assert r.returncode == 0
    assert (tmp_path / "a.fa").read_text() == ">fwd\n%s\n>rev\n%s\n>edge\n%s\n" % (seq[100:123], revcomp(seq[200:223]), seq[2:25])
    # '+': 4 upstream + 3 downstream; '-': 3 before + 4 after on the forward strand, then revcomp;
    # start 2 - 4 wraps (unsigned) and is clamped to the contig length -> empty sequence, as in the reference
    assert (tmp_path / "b.fa").read_text() == ">fwd\n%s\n>rev\n%s\n>edge\n\n" % (seq[96:126], revcomp(seq[197:227]))
    assert run("fasta_writer", "a", "b").returncode == 1


def test_driver_reports_a_vcf_without_sample_columns(tmp_path):
    """`-s all` on a VCF without a #CHROM line (or without sample columns): the driver's own message and exit code 2
    (usage error, VARSCOT:213-249) - under `set -Eeuo pipefail` a failing `grep | awk` pipeline used to end the
    script on the assignment, with exit code 1 and no text.  Needs no device: the check comes before any tool."""
    import subprocess
    driver = os.path.join(ROOT, "varscot_amd", "driver", "VARSCOT")
    for body in ("##fileformat=VCFv4.2\n", "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\n"):
        vcf = tmp_path / "no_samples.vcf"
        vcf.write_text(body)
        r = subprocess.run(["bash", driver, "-f", str(vcf), "-s", "all", "-b", "x.bed", "-o", "out.txt", "-g", "g.fa", "-i", "ix",
                            "-T", str(tmp_path / "tmp")], capture_output=True, text=True)
        assert r.returncode == 2, (r.returncode, r.stdout, r.stderr)
        assert "Error: The VCF file has no sample columns." in r.stdout


def test_bidir_index_removes_a_stale_seed_index_file(tmp_path):
    """`bidir_index` without -S rewrites <prefix>.vsc; a <prefix>.vsi left from an earlier genome under the same
    prefix would be auto-loaded by bidir_mapping, so it goes.  Host-only (packing needs no device)."""
    rng = np.random.default_rng(3)
    fa = str(tmp_path / "g.fa")
    write_fasta(fa, ["chr1"], ["".join(rng.choice(list("ACGT"), size=500))])
    prefix = str(tmp_path / "ix")
    open(prefix + ".vsi", "wb").write(b"stale")
    r = run("bidir_index", "-G", fa, "-I", prefix)
    assert r.returncode == 0 and "Index created successfully" in r.stdout
    assert os.path.exists(prefix + ".vsc") and not os.path.exists(prefix + ".vsi")
