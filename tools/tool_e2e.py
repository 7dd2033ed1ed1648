import sys, os, time, subprocess, tempfile
sys.path.insert(0, os.getcwd())
from varscot_amd import synth
tmp = tempfile.mkdtemp(prefix="e2e_")
t = time.time(); full = synth.synthetic_genome(3_000_000_000); synth.write_fasta(full, tmp + "/genome.fa"); print("genome fasta %.1f s" % (time.time() - t), flush=True)
ids, seqs = synth.synthetic_guides(1000)
with open(tmp + "/reads.fa", "w") as f:
    for i, s in zip(ids, seqs):
        f.write(">%s\n%s\n" % (i, s))
b = "varscot_amd/bin/"
t = time.time(); subprocess.check_call([b + "bidir_index", "-G", tmp + "/genome.fa", "-I", tmp + "/idx"], stdout=subprocess.DEVNULL); print("bidir_index %.1f s" % (time.time() - t), flush=True)
t = time.time(); subprocess.check_call([b + "bidir_mapping", "-G", tmp + "/genome.fa", "-I", tmp + "/idx", "-R", tmp + "/reads.fa", "-M", "6", "-O", tmp + "/out.sam"], stdout=subprocess.DEVNULL); print("bidir_mapping %.1f s, SAM %.0f MB" % (time.time() - t, os.path.getsize(tmp + "/out.sam") / 1e6), flush=True)
import shutil; shutil.rmtree(tmp)
