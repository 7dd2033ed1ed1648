#!/bin/bash
# driver_full.sh ref|vcf - the VARSCOT driver at a realistic size on the GPU box: 16 on-targets on the 3 Gbp synthetic genome,
#   ref: <= 8 mismatches, -e prob, no VCF (2.6 M off-targets + their 443-column feature matrix)
#   vcf: <= 6 mismatches, MIT scores, 5 M-SNP VCF (one sample)
# Three runs each: the one-process route cold (no <prefix>.vsc yet: the FASTA is packed first), the same warm, and the staged
# route (VARSCOT_STAGED=1) - whose result files must be byte-identical.  Stage times -> gpurun_out/<TAG>/.
set -e
MODE=${1:?ref|vcf}; TAG=${TAG:-driver_full}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p "$OUT"
D=/tmp/vsc_drv_$MODE; rm -rf $D; mkdir -p $D
python3 - "$D" "$MODE" <<'PY'
import sys
sys.path.insert(0, ".")
from varscot_amd import synth, _lib
D, mode = sys.argv[1], sys.argv[2]
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
synth.write_fasta(packed, D + "/genome.fa")
if mode == "vcf":
    print("snps", synth.synthetic_vcf(packed, 5_000_000, D + "/in.vcf"))
open(D + "/targets.bed", "w").write("\n".join(bed) + "\n")
open(D + "/activity.txt", "w").write("\n".join(act) + "\n")
PY
if [ "$MODE" == ref ]; then ARGS=(-m 8 -e prob); else ARGS=(-m 6 -f $D/in.vcf -s 0); fi
one() {  # one NAME [ENV=VALUE]
  local name=$1; shift
  local t0=$(date +%s%N)
  env "$@" VARSCOT_TRACE=1 PS4='+ $(date +%s.%N) ' bash -x varscot_amd/driver/VARSCOT -b $D/targets.bed -o $D/$name.txt -g $D/genome.fa -i $D/idx -t 16 \
      -T $D/tmp_$name -a $D/activity.txt "${ARGS[@]}" > "$OUT/${MODE}_$name.stdout" 2> "$OUT/${MODE}_$name.trace"
  local t1=$(date +%s%N)
  echo "$MODE $name: $(( (t1 - t0) / 1000000 )) ms, $(wc -l < $D/$name.txt) lines" | tee -a "$OUT/${MODE}_summary.txt"
  grep -E "^\[varscot_pipeline\]" "$OUT/${MODE}_$name.trace" | tee -a "$OUT/${MODE}_summary.txt" || true
}
: > "$OUT/${MODE}_summary.txt"
one cold
one warm
one staged VARSCOT_STAGED=1
cmp $D/warm.txt $D/staged.txt && cmp $D/cold.txt $D/staged.txt && echo "$MODE: result files identical" | tee -a "$OUT/${MODE}_summary.txt"
if [ "$MODE" == ref ]; then cmp $D/warm_feature_matrix.txt $D/staged_feature_matrix.txt && echo "ref: feature matrices identical ($(du -h $D/warm_feature_matrix.txt | cut -f1))" | tee -a "$OUT/${MODE}_summary.txt"; fi
# the staged route's stages, from the trace's time stamps
python3 - "$OUT/${MODE}_staged.trace" <<'PY' | tee -a "$OUT/${MODE}_summary.txt"
import re, sys
rows = []
for line in open(sys.argv[1], errors="replace"):
    m = re.match(r"\++ (\d+\.\d+) (.*)", line)
    if m:
        rows.append((float(m.group(1)), m.group(2)))
for (t, cmd), (t2, _) in zip(rows, rows[1:]):
    if t2 - t > 0.3:
        print("staged  %7.2f s  %s" % (t2 - t, cmd[:110]))
PY
rm -rf $D
