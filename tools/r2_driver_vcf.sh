#!/bin/bash
# the VARSCOT driver with a VCF at full size: 16 on-targets, 3 Gbp genome, 5 M SNPs, <= 6 mismatches
set -e
mkdir -p gpurun_out/r2c
D=/tmp/vsc_drv_vcf; rm -rf $D; mkdir -p $D
python - <<PY
import sys
sys.path.insert(0, ".")
from varscot_amd import synth, _lib
packed = synth.synthetic_genome(3_000_000_000)
ids, guides = synth.synthetic_guides(16)
L = _lib.lib()
bed, act = [], ["ID   Sequence   Score   Dir"]
for i, g in enumerate(guides):
    c = i % len(packed.contigs)
    pos = 5_000_000 + 1_000_003 * i
    L.vsc_pack_bases(g.encode(), 23, int(packed.contigs[c]["offset"]) + pos, _lib.ptr(packed.hi), _lib.ptr(packed.lo), _lib.ptr(packed.nmask))
    bed.append("%s\t%d\t%d\tsite%d\t7\t+" % (packed.names[c], pos, pos + 23, i))
    act.append("site%d %s %.6f +" % (i, "A" * 30, 0.3 + 0.04 * i))
synth.plant_sites(packed, guides, 2000, 5)
synth.write_fasta(packed, "$D/genome.fa")
print("snps", synth.synthetic_vcf(packed, 5_000_000, "$D/in.vcf"))
open("$D/targets.bed", "w").write("\n".join(bed) + "\n")
open("$D/activity.txt", "w").write("\n".join(act) + "\n")
PY
T0=$(date +%s%N)
PS4="+ \$(date +%s.%N) " bash -x varscot_amd/driver/VARSCOT -b $D/targets.bed -o $D/result.txt -g $D/genome.fa -i $D/idx -m 6 -t 16 -T $D/tmp -a $D/activity.txt -f $D/in.vcf -s 0 2> gpurun_out/r2c/driver_vcf_trace.txt | tail -5
T1=$(date +%s%N); echo "driver $(( (T1 - T0) / 1000000 )) ms"
wc -l $D/result.txt; cut -f 10 $D/result.txt | sort | uniq -c | sort -rn | head -5
rm -rf $D
