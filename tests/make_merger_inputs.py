#!/usr/bin/env python3
"""Inputs for the mergers' sanitizer run (tools/multi_tsan/run.sh): the scenario of tests/test_pipeline.py written into
DIR, the SNP genome made by the real vcf_loader, the two SAM files taken from the ORACLE's search (test infrastructure: the
product's search needs the GPU), and a packed genome of each FASTA.  Lives under tests/ because it calls the oracle (only
tests/, smoke() and bench.py's cpu_baseline may).   usage: tests/make_merger_inputs.py DIR SEED"""
import os
import pathlib
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
from test_pipeline import BIN, build_scenario, read_fasta  # noqa: E402

d = pathlib.Path(sys.argv[1])
d.mkdir(parents=True, exist_ok=True)
pyoracle.build()
d, records, bed, tus, targets = build_scenario(d, int(sys.argv[2]))
subprocess.run([os.path.join(BIN, "vcf_loader"), str(d / "in.vcf"), str(d / "snp.fa"), str(d / "genome.fa"), "0", "23", "2"], check=True,
               stdout=subprocess.DEVNULL)
for fasta, recs, out in (("genome.fa", records, "ref.sam"), ("snp.fa", read_fasta(d / "snp.fa"), "snp.sam")):
    sam = pyoracle.search_sam([s for _, s in recs], [n for n, _ in recs], [t[4] for t in targets], [t[0] for t in targets], 5, None, 0)
    (d / out).write_text(sam)
    subprocess.run([os.path.join(BIN, "bidir_index"), "-G", str(d / fasta), "-I", str(d / fasta.replace(".fa", "_idx"))], check=True,
                   stdout=subprocess.DEVNULL)
print(len((d / "ref.sam").read_text().splitlines()), len((d / "snp.sam").read_text().splitlines()))
