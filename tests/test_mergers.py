"""The host stages behind the search - vcf_loader, bam_merger_ref_only, bam_merger - fed with the ORACLE's SAM text for the
scenario's genomes (oracle.search_sam, what bidir_mapping's output is compared with elsewhere) instead of the search's own:
their TSV is compared byte for byte with the restatement of variant_processing/{merge_output_bam,filter_output_bam}.h
(oracle/merge_oracle.py) on three scenarios, and a SAM file with a broken record ends the way the reference's does.
(The mergers score every row on the device - there is no CPU fallback -, so these are GPU tests.)"""
import os
import subprocess

import pytest

from oracle import merge_oracle as mo
from oracle import variants_oracle as vo
from test_pipeline import BIN, build_scenario, read_fasta, run


def oracle_sam(oracle, records, targets, max_mm):
    names = [n for n, _ in records]
    return oracle.search_sam([s for _, s in records], names, [g for *_x, g in targets], [t[0] for t in targets], max_mm, None, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [20240, 7, 99])
def test_mergers_on_the_oracles_sam_text(tmp_path, oracle, seed):
    d, records, bed, tus, targets = build_scenario(tmp_path, seed)
    run("vcf_loader", d / "in.vcf", d / "snp.fa", d / "genome.fa", 0, 23, 2)
    assert (d / "snp.fa").read_text() == vo.format_fasta(vo.vcf_loader((d / "in.vcf").read_text(), dict(records), 0, 23))
    snp_records = read_fasta(d / "snp.fa")
    ref_sam = oracle_sam(oracle, records, targets, 5)
    snp_sam = oracle_sam(oracle, snp_records, targets, 5)
    assert len(ref_sam.splitlines()) > 30 and len(snp_sam.splitlines()) > 10
    (d / "ref.sam").write_text(ref_sam)
    (d / "snp.sam").write_text(snp_sam)
    run("bam_merger_ref_only", d / "out.txt", d / "feat.txt", d / "ref.sam", d / "targets.bed", d / "genome.fa", d / "activity.txt", 5, 23, 0)
    assert (d / "out.txt").read_text() == mo.process_ref_only(ref_sam, bed, records, tus, False)[0]
    run("bam_merger", d / "merged.txt", d / "mfeat.txt", d / "ref.sam", d / "snp.sam", d / "targets.bed", d / "genome.fa", d / "snp.fa",
        d / "activity.txt", 5, 23, 2, 0)
    want = mo.merge_results(ref_sam, snp_sam, bed, records, snp_records, tus, 23, False)[0]
    assert (d / "merged.txt").read_text() == want
    rows = want.splitlines()[1:]
    assert any(r.split("\t")[-1].startswith("VAR_") for r in rows) and any(r.split("\t")[-1] == "REF" for r in rows)


@pytest.mark.gpu
def test_merger_stops_at_a_record_it_cannot_parse(tmp_path, oracle):
    """A SAM file cut in the middle of a record: the reference catches SeqAn's exception, prints it and carries on with the
    records read before it (filter_output_bam.h:413-417) - so do the drop-ins: a line on stdout, the table of the intact records."""
    d, records, bed, tus, targets = build_scenario(tmp_path)
    sam = oracle_sam(oracle, records, targets, 4)
    lines = sam.splitlines()
    k = len(lines) // 2
    cut = "\n".join(lines[:k] + ["\t".join(lines[k].split("\t")[:7])] + lines[k + 1:]) + "\n"  # record k loses its last fields
    (d / "ref.sam").write_text(cut)
    r = subprocess.run([os.path.join(BIN, "bam_merger_ref_only"), str(d / "out.txt"), str(d / "feat.txt"), str(d / "ref.sam"),
                        str(d / "targets.bed"), str(d / "genome.fa"), str(d / "activity.txt"), "4", "23", "1"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "line %d" % (k + 1) in r.stdout
    want = mo.process_ref_only("\n".join(lines[:k]) + "\n", bed, records, tus, True)[0]
    assert (d / "out.txt").read_text() == want == mo.process_ref_only(cut, bed, records, tus, True)[0]
    assert len(want.splitlines()) > 5
