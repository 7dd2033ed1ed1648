"""The mergers' own host code on the CPU: bam_merger_ref_only / bam_merger (tools/merge_host.hpp) built with the scoring calls
answered by tools/multi_tsan/stub_scores.cpp (INVENTED scores - the real ones are computed on the device, there is no CPU
path) and fed with the oracle's SAM text: every column of the result except the score - locus, name and running number,
strand, sequence, mismatch number and positions, variant annotation - and the row order must equal the restatement of
variant_processing/{merge_output_bam,filter_output_bam}.h (oracle/merge_oracle.py).  The scores themselves, the feature
matrix and the forest are tests/test_pipeline.py's and tests/test_mergers.py's (GPU)."""
import os
import shutil
import subprocess

import pytest

from oracle import merge_oracle as mo
from test_pipeline import BIN, build_scenario, read_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "varscot_amd", "csrc")
STUBS = os.path.join(ROOT, "tools", "multi_tsan")


@pytest.fixture(scope="module")
def stand_in_mergers(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    d = tmp_path_factory.mktemp("stand_in")
    for tool in ("bam_merger_ref_only", "bam_merger"):
        b = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + STUBS,
                            os.path.join(CSRC, "tools", tool + ".cpp"), os.path.join(STUBS, "stub_scores.cpp"), os.path.join(CSRC, "vsc_pack.cpp"),
                            "-pthread", "-o", str(d / tool)], capture_output=True, text=True, timeout=600)
        assert b.returncode == 0, b.stderr[-2000:]
    return d


def without_scores(tsv):
    rows = [line.split("\t") for line in tsv.splitlines()]
    assert rows[0][4] == "Score"
    return [r[:4] + r[5:] for r in rows]


@pytest.mark.parametrize("seed,packed", [(20240, False), (7, True), (99, False)])
def test_mergers_rows_without_the_scores_equal_the_restatement(tmp_path, oracle, stand_in_mergers, seed, packed):
    d, records, bed, tus, targets = build_scenario(tmp_path, seed)
    r = subprocess.run([os.path.join(BIN, "vcf_loader"), str(d / "in.vcf"), str(d / "snp.fa"), str(d / "genome.fa"), "0", "23", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    snp_records = read_fasta(d / "snp.fa")

    def sam_of(recs):
        return oracle.search_sam([s for _, s in recs], [n for n, _ in recs], [t[4] for t in targets], [t[0] for t in targets], 5, None, 0)

    ref_sam, snp_sam = sam_of(records), sam_of(snp_records)
    (d / "ref.sam").write_text(ref_sam)
    (d / "snp.sam").write_text(snp_sam)
    env = dict(os.environ, VARSCOT_TRACE="1")
    if packed:  # reference bases from the packed genomes instead of the FASTA text
        for fasta, prefix, var in (("genome.fa", "ref_idx", "VARSCOT_PACKED_GENOME"), ("snp.fa", "snp_idx", "VARSCOT_PACKED_SNP_GENOME")):
            assert subprocess.run([os.path.join(BIN, "bidir_index"), "-G", str(d / fasta), "-I", str(d / prefix)], capture_output=True).returncode == 0
            env[var] = str(d / prefix)
    r = subprocess.run([str(stand_in_mergers / "bam_merger_ref_only"), str(d / "out.txt"), str(d / "feat.txt"), str(d / "ref.sam"), str(d / "targets.bed"),
                        str(d / "genome.fa"), str(d / "activity.txt"), "5", "23", "0"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and ("packed genome" in r.stderr) == packed, r.stdout + r.stderr
    want = mo.process_ref_only(ref_sam, bed, records, tus, False)[0]
    assert without_scores((d / "out.txt").read_text()) == without_scores(want) and len(want.splitlines()) > 20
    r = subprocess.run([str(stand_in_mergers / "bam_merger"), str(d / "merged.txt"), str(d / "mfeat.txt"), str(d / "ref.sam"), str(d / "snp.sam"),
                        str(d / "targets.bed"), str(d / "genome.fa"), str(d / "snp.fa"), str(d / "activity.txt"), "5", "23", "2", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    want = mo.merge_results(ref_sam, snp_sam, bed, records, snp_records, tus, 23, False)[0]
    assert without_scores((d / "merged.txt").read_text()) == without_scores(want)
    assert any(row[-1].startswith("VAR_") for row in without_scores(want)[1:])


def test_the_shipped_mergers_have_no_cpu_path(tmp_path, oracle):
    """The stand-in above is test scaffolding: the mergers as they are SHIPPED (varscot_amd/bin) score on the device and say so when
    there is none - they do not fall back to anything.  (Skipped where a device is visible: tests/test_pipeline.py runs them there.)"""
    import varscot_amd as va
    if va.device_count() > 0:
        pytest.skip("a HIP device is visible")
    d, records, bed, tus, targets = build_scenario(tmp_path)
    sam = oracle.search_sam([s for _, s in records], [n for n, _ in records], [t[4] for t in targets], [t[0] for t in targets], 4, None, 0)
    (d / "ref.sam").write_text(sam)
    r = subprocess.run([os.path.join(BIN, "bam_merger_ref_only"), str(d / "out.txt"), str(d / "feat.txt"), str(d / "ref.sam"), str(d / "targets.bed"),
                        str(d / "genome.fa"), str(d / "activity.txt"), "4", "23", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device available (there is no CPU fallback)" in r.stdout
    # (and nothing of the stand-ins is part of the library or the tools)
    makefile = open(os.path.join(CSRC, "Makefile")).read()
    assert "multi_tsan" not in makefile and "stub_" not in makefile
