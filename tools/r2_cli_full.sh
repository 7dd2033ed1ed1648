#!/bin/bash
# the command-line drop-ins at full size: 3 Gbp FASTA -> bidir_index -> bidir_mapping (1 000 reads, <= 6 mismatches) -> SAM
set -e
mkdir -p gpurun_out/r2c
D=/tmp/vsc_cli_full; rm -rf $D; mkdir -p $D
python - <<PY
import time, sys
sys.path.insert(0, ".")
from varscot_amd import synth
t=time.time(); packed = synth.synthetic_genome(3_000_000_000); print("genome", round(time.time()-t,1), "s")
ids, guides = synth.synthetic_guides(1000)
synth.plant_sites(packed, guides, 400, 6)
t=time.time(); synth.write_fasta(packed, "$D/genome.fa"); print("write fasta", round(time.time()-t,1), "s")
open("$D/reads.fa","w").write("".join(">%s\n%s\n" % (i, g) for i, g in zip(ids, guides)))
PY
ls -la $D
T0=$(date +%s%N)
varscot_amd/bin/bidir_index -G $D/genome.fa -I $D/idx | tail -2
T1=$(date +%s%N); echo "bidir_index $(( (T1 - T0) / 1000000 )) ms"
varscot_amd/bin/bidir_mapping -G $D/genome.fa -I $D/idx -R $D/reads.fa -M 6 -T 16 -O $D/out.sam | tail -2
T2=$(date +%s%N); echo "bidir_mapping $(( (T2 - T1) / 1000000 )) ms"
wc -l $D/out.sam; head -2 $D/out.sam; md5sum $D/out.sam
rm -rf $D
