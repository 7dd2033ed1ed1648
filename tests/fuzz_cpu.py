#!/usr/bin/env python3
"""Randomised sweeps of the HOST tools on the CPU (not collected by pytest; the GPU counterpart is tests/fuzz_gpu.py):

    python tests/fuzz_cpu.py variants [N]   random genomes (1-4 contigs of 120-5 000 bases, N runs) and VCFs (SNPs, indels, multi-allelic,
                                            phased / unphased, clustered, both sample columns): vcf_loader == the restatement of
                                            variant_processing/{process_vcf,overlap_sequences,write_fasta}.h (oracle/variants_oracle.py),
                                            and vsc_windows_build (planes, contig table, ids) == vcf_loader + packing, on 0-5 threads
    python tests/fuzz_cpu.py ontargets [N]  random BED6 records (starts up to past the contig end, lengths 5 / 23 / 40, both strands,
                                            soft-masked and IUPAC stretches, several line widths): fasta_writer's two files == a
                                            restatement of extract_fasta_ontargets.h:33-76 written out here, from the FASTA text
                                            and from the packed genome
    python tests/fuzz_cpu.py mergers [N]    the pipeline tests' scenario under N seeds: bam_merger_ref_only / bam_merger (built over invented
                                            scores: their scoring runs on the device) on the oracle's SAM text - every column but the
                                            score == the restatement of variant_processing/{merge_output_bam,filter_output_bam}.h
    python tests/fuzz_cpu.py routes [N]     driver/VARSCOT over stand-in builds of the device-bound tools: the one-process route's files ==
                                            the staged route's, byte for byte, under N scenario seeds (mit / prob / class, one or all samples)
"""
import os
import pathlib
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_seq, revcomp  # noqa: E402

BIN = os.path.join(ROOT, "varscot_amd", "bin")


def variants(n):
    import test_variants as tv
    import varscot_amd as va
    vo = tv.vo
    t0, fails, refused = time.time(), 0, 0
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="vsc_fuzz_"))
    for seed in range(100, 100 + n):
        rng = np.random.default_rng(5000 + seed)
        sizes = [int(rng.integers(120, 5000)) for _ in range(int(rng.integers(1, 5)))]
        genome = {"c%d" % i: random_seq(rng, s) for i, s in enumerate(sizes)}
        if len(genome["c0"]) > 200 and seed % 3 == 0:
            a, run = int(rng.integers(0, len(genome["c0"]) - 60)), int(rng.integers(1, 50))
            genome["c0"] = genome["c0"][:a] + "N" * run + genome["c0"][a + run:]
        sample, indel = int(rng.integers(0, 2)), float(rng.choice([0.0, 0.2, 0.5, 0.9]))
        n_rec = int(rng.integers(1, max(2, min(400, sum(sizes) // 12))))  # (denser VCFs: 2^n windows per range, see below)
        for f in tmp.iterdir():
            f.unlink()
        vcf = tv.synth_vcf(seed, genome, n_rec, header_contigs=None, indel_rate=indel, cluster=bool(seed % 2))
        r, got = tv.run_tool(tmp, vcf, genome, sample)
        if r.returncode == 1 and "unphased variants within one window" in r.stdout:
            refused += 1  # more than 24 unphased records in one range: refused by design (tools/vcf_expand.hpp)
            continue
        ok = r.returncode == 0 and got == vo.format_fasta(vo.vcf_loader(vcf, genome, sample, 23))
        if ok and got.count(">"):
            want = tv._fasta_route(tmp, vcf, genome, sample)
            ref = va.PackedGenome.from_sequences(list(genome.values()), [name + " some description" for name in genome])
            g = va.variant_windows(ref, tmp / "in.vcf", sample=sample, threads=int(rng.integers(0, 6)))
            ok = (g.contigs.tobytes() == want.contigs.tobytes() and list(g.names) == want.names and
                  all(a.tobytes() == b.tobytes() for a, b in ((g.hi, want.hi), (g.lo, want.lo), (g.nmask, want.nmask))))
        if not ok:
            fails += 1
            print("FAIL seed", seed, sizes, sample, indel, n_rec, flush=True)
        if (seed - 99) % 250 == 0:
            print("%d / %d configurations, %d failures, %d refused, %d s" % (seed - 99, n, fails, refused, time.time() - t0), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print("variants fuzz: %d configurations, %d failures" % (n, fails))
    return fails


def ontargets(n):
    def dna5(s):
        return "".join(c if c in "ACGT" else "N" for c in s.upper())

    def extract(genome, chrom, start, end, strand, flank):  # extract_fasta_ontargets.h:33-76 (unsigned arithmetic, clamps)
        seq = genome[chrom]
        if flank:
            start, end = (start - 4, end + 3) if strand == "+" else (start - 3, end + 4)
        start, end = start & 0xFFFFFFFF, end & 0xFFFFFFFF
        start, end = min(start, len(seq)), min(end, len(seq))
        end = max(end, start)
        s = dna5(seq[start:end])
        return revcomp(s) if strand == "-" else s

    d = pathlib.Path(tempfile.mkdtemp(prefix="vsc_fuzz_"))
    fails = 0
    for seed in range(n):
        rng = np.random.default_rng(9000 + seed)
        names = ["c%d desc" % i for i in range(int(rng.integers(1, 5)))]
        genome = {}
        for name in names:
            s = random_seq(rng, int(rng.integers(1, 400)))
            if rng.random() < 0.5 and len(s) > 20:
                a = int(rng.integers(0, len(s) - 10))
                s = s[:a] + s[a:a + 6].lower() + "NRY" + s[a + 9:]
            genome[name.split()[0]] = s
        with open(d / "g.fa", "w") as f:
            for name in names:
                s, w = genome[name.split()[0]], int(rng.choice([10, 60, 61]))
                f.write(">%s\n" % name)
                for i in range(0, len(s), w):
                    f.write(s[i:i + w] + "\n")
        bed = []
        for k in range(int(rng.integers(1, 30))):
            c = names[int(rng.integers(0, len(names)))].split()[0]
            start = int(rng.integers(0, len(genome[c]) + 6))
            bed.append((c, start, start + int(rng.choice([23, 23, 23, 5, 40])), "t%d" % k, "+-"[int(rng.integers(0, 2))]))
        (d / "t.bed").write_text("".join("%s\t%d\t%d\t%s\t0\t%s\n" % b for b in bed))
        want = ["".join(">%s\n%s\n" % (name, extract(genome, c, s, e, st, flank)) for c, s, e, name, st in bed) for flank in (False, True)]
        for mode in ("text", "packed"):
            for name in ("g.fa.vsc", "a.fa", "b.fa"):
                if (d / name).exists():
                    (d / name).unlink()
            if mode == "packed":
                subprocess.run([os.path.join(BIN, "bidir_index"), "-G", str(d / "g.fa"), "-I", str(d / "g.fa")], check=True, capture_output=True)
            r = subprocess.run([os.path.join(BIN, "fasta_writer"), str(d / "a.fa"), str(d / "b.fa"), str(d / "t.bed"), str(d / "g.fa")],
                               capture_output=True, text=True, env=dict(os.environ, VARSCOT_TRACE="1"))
            if not (r.returncode == 0 and (d / "a.fa").read_text() == want[0] and (d / "b.fa").read_text() == want[1] and
                    ("packed genome" in r.stderr) == (mode == "packed")):
                fails += 1
                print("FAIL seed", seed, mode, r.returncode, r.stdout[-200:], r.stderr[-200:], flush=True)
                break
    shutil.rmtree(d, ignore_errors=True)
    print("ontargets fuzz: %d configurations (FASTA text and packed genome), %d failures" % (n, fails))
    return fails


def mergers(n):
    """Scenarios of tests/test_pipeline.py under n seeds (on-targets, planted off-targets, SNPs / indels next to them), the oracle's
    SAM text for the genome and for the SNP genome of the real vcf_loader, the two mergers built over invented scores
    (tools/multi_tsan/stub_scores.cpp - their scoring runs on the device): every column but the score == oracle/merge_oracle.py."""
    from oracle import merge_oracle as mo
    from oracle import pyoracle
    from test_pipeline import build_scenario, read_fasta
    pyoracle.build()
    csrc, stubs = os.path.join(ROOT, "varscot_amd", "csrc"), os.path.join(ROOT, "tools", "multi_tsan")
    bin_dir = pathlib.Path(tempfile.mkdtemp(prefix="vsc_fuzz_bin_"))
    for tool in ("bam_merger_ref_only", "bam_merger"):
        subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-I" + stubs, os.path.join(csrc, "tools", tool + ".cpp"),
                        os.path.join(stubs, "stub_scores.cpp"), os.path.join(csrc, "vsc_pack.cpp"), "-pthread", "-o", str(bin_dir / tool)], check=True)

    def cut(tsv):
        return [r.split("\t")[:4] + r.split("\t")[5:] for r in tsv.splitlines()]
    fails, t0 = 0, time.time()
    for seed in range(1, n + 1):
        d = pathlib.Path(tempfile.mkdtemp(prefix="vsc_fuzz_"))
        d, records, bed, tus, targets = build_scenario(d, seed)
        if seed % 2:  # every other scenario with a denser, wilder VCF: indels, multi-allelic and unphased records, clusters (tests/test_variants.py)
            import test_variants as tv
            rng = np.random.default_rng(seed)
            (d / "in.vcf").write_text(tv.synth_vcf(seed, dict(records), int(rng.integers(40, 500)), n_samples=1, indel_rate=float(rng.choice([0.0, 0.3, 0.7])),
                                                   cluster=bool(rng.integers(0, 2))))
        vr = subprocess.run([os.path.join(BIN, "vcf_loader"), str(d / "in.vcf"), str(d / "snp.fa"), str(d / "genome.fa"), "0", "23", "2"], capture_output=True, text=True)
        if vr.returncode == 1 and "unphased variants within one window" in vr.stdout:
            shutil.rmtree(d, ignore_errors=True)
            continue
        assert vr.returncode == 0, vr.stdout
        snp_records = read_fasta(d / "snp.fa")
        mm = 3 + seed % 3
        sams = []
        for recs, name in ((records, "ref.sam"), (snp_records, "snp.sam")):
            sams.append(pyoracle.search_sam([s for _, s in recs], [x for x, _ in recs], [t[4] for t in targets], [t[0] for t in targets], mm, None, 0))
            (d / name).write_text(sams[-1])
        r1 = subprocess.run([str(bin_dir / "bam_merger_ref_only"), str(d / "o1.txt"), str(d / "f1.txt"), str(d / "ref.sam"), str(d / "targets.bed"), str(d / "genome.fa"),
                             str(d / "activity.txt"), str(mm), "23", "0"], capture_output=True, text=True)
        r2 = subprocess.run([str(bin_dir / "bam_merger"), str(d / "o2.txt"), str(d / "f2.txt"), str(d / "ref.sam"), str(d / "snp.sam"), str(d / "targets.bed"),
                             str(d / "genome.fa"), str(d / "snp.fa"), str(d / "activity.txt"), str(mm), "23", "2", "0"], capture_output=True, text=True)
        ok = (r1.returncode == 0 and r2.returncode == 0 and
              cut((d / "o1.txt").read_text()) == cut(mo.process_ref_only(sams[0], bed, records, tus, False)[0]) and
              cut((d / "o2.txt").read_text()) == cut(mo.merge_results(sams[0], sams[1], bed, records, snp_records, tus, 23, False)[0]))
        if not ok:
            fails += 1
            print("FAIL seed", seed, r1.returncode, r2.returncode, flush=True)
        shutil.rmtree(d, ignore_errors=True)
        if seed % 50 == 0:
            print("%d / %d scenarios, %d failures, %d s" % (seed, n, fails, time.time() - t0), flush=True)
    shutil.rmtree(bin_dir, ignore_errors=True)
    print("mergers fuzz: %d scenarios, %d failures" % (n, fails))
    return fails


def routes(n):
    """The driver's one-process route against its staged route (tests/test_driver_stand_in.py) under n scenario seeds, every other one
    with a wild synthetic VCF: both over the stand-in builds of the device-bound tools, result files byte for byte."""
    import test_driver_stand_in as ts
    from test_pipeline import build_scenario

    class Factory:
        def mktemp(self, name):
            return pathlib.Path(tempfile.mkdtemp(prefix="vsc_fuzz_" + name))
    bin_dir = ts.stand_in_bin.__wrapped__(Factory()) if hasattr(ts.stand_in_bin, "__wrapped__") else None
    driver = os.path.join(ROOT, "varscot_amd", "driver", "VARSCOT")
    fails, t0 = 0, time.time()
    for seed in range(1, n + 1):
        tmp = pathlib.Path(tempfile.mkdtemp(prefix="vsc_fuzz_"))
        d, records, bed, tus, targets = build_scenario(tmp, seed)
        if seed % 2:
            import test_variants as tv
            rng = np.random.default_rng(seed)
            (d / "in.vcf").write_text(tv.synth_vcf(seed, dict(records), int(rng.integers(40, 500)), n_samples=2, indel_rate=float(rng.choice([0.0, 0.3, 0.7])),
                                                   cluster=bool(rng.integers(0, 2))))
        evaluation = ["mit", "prob", "class"][seed % 3]
        got = {}
        for route in ("inproc", "staged"):
            out = tmp / ("res_%s.txt" % route)
            cmd = ["bash", driver, "-b", str(d / "targets.bed"), "-o", str(out), "-g", str(d / "genome.fa"), "-i", str(tmp / "idx"), "-m", str(3 + seed % 3),
                   "-t", "2", "-T", str(tmp / ("tmp_" + route)), "-a", str(d / "activity.txt"), "-e", evaluation, "-f", str(d / "in.vcf"),
                   "-s", "all" if seed % 2 else "0"]
            env = dict(os.environ, VARSCOT_BIN=str(bin_dir), VARSCOT_RF_MODEL=os.path.join(ROOT, "varscot_amd", "models", "rfClassifier.vscrf"))
            if route == "staged":
                env["VARSCOT_STAGED"] = "1"
            r = subprocess.run(cmd, capture_output=True, text=True, env=env)
            files = {p.name.replace(route, ""): p.read_bytes() for p in sorted(tmp.glob("res_%s*.txt" % route))}
            got[route] = (r.returncode, files)
        refused = any("unphased variants within one window" in str(v) for v in got.values())
        if got["inproc"] != got["staged"] or (got["inproc"][0] != 0 and not refused) or (got["inproc"][0] == 0 and not got["inproc"][1]):
            fails += 1
            print("FAIL seed", seed, evaluation, got["inproc"][0], got["staged"][0], sorted(got["inproc"][1]), sorted(got["staged"][1]), flush=True)
        shutil.rmtree(tmp, ignore_errors=True)
        if seed % 25 == 0:
            print("%d / %d scenarios, %d failures, %d s" % (seed, n, fails, time.time() - t0), flush=True)
    shutil.rmtree(bin_dir, ignore_errors=True)
    print("routes fuzz: %d scenarios, %d failures" % (n, fails))
    return fails


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "variants"
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    sys.exit(1 if {"variants": variants, "ontargets": ontargets, "mergers": mergers, "routes": routes}[what](count) else 0)
