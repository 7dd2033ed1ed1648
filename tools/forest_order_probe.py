import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import varscot_amd as va
from varscot_amd import synth
from varscot_amd.classifier import Forest
ctx = va.Context(0)
packed = synth.synthetic_genome(3_000_000_000)
ids, guides = synth.synthetic_guides(10_000)
genome = ctx.load_genome(packed); genome.build_index()
forest = Forest()
h = genome.search(guides[:400], 8, algorithm="seed")
n = len(h); print("hits", n)
rows = torch.empty((n, 16), dtype=torch.int32, device="cuda:0")
h.packed_features(to_host=False, dev_ptr=rows.data_ptr())
from varscot_amd.dist import DeviceAlias
rec = torch.as_tensor(DeviceAlias(h.device_ptr, n * 16), device="cuda:0").view(torch.int32).view(-1, 4)
g = rec[:, 0].cpu().numpy()
act_g = np.random.default_rng(5).uniform(0.2, 1.8, size=400)
act = act_g[g]
def run(r, a, label):
    torch.cuda.synchronize()
    for _ in range(2):
        prob, cls, tie = forest.predict_packed(ctx, len(a), a, dev_ptr=r.data_ptr())
    print(label, "score_ms", round(ctx.timing()["score_ms"], 2), "mean prob", float(prob.mean()))
    return prob
p0 = run(rows, act, "natural order (read, strand, position)")
# within each read: rows ordered by their mismatch-flag word (w0 low 21 bits), then by the type flags
key = (rec[:, 0].to(torch.int64) << 40) | ((rows[:, 0].to(torch.int64) & 0x1FFFFF) << 12) | (rows[:, 1].to(torch.int64) & 0xFFF)
order = torch.argsort(key)
rows2 = rows[order].contiguous()
act2 = act[order.cpu().numpy()]
p1 = run(rows2, act2, "within a read by mismatch positions, types")
assert np.allclose(np.sort(p0), np.sort(p1))
# by the full set of off-target one-hot words too
key3 = (rec[:, 0].to(torch.int64) << 42) | ((rows[:, 0].to(torch.int64) & 0x1FFFFF) << 21) | ((rows[:, 2].to(torch.int64) ^ rows[:, 3].to(torch.int64) ^ rows[:, 4].to(torch.int64)) & 0x1FFFFF)
order = torch.argsort(key3)
rows3 = rows[order].contiguous(); act3 = act[order.cpu().numpy()]
run(rows3, act3, "by mismatch positions, then a hash of the base one-hots")
# random order inside a read (how much coherence the natural order has)
key4 = (rec[:, 0].to(torch.int64) << 40) | torch.randint(0, 1 << 39, (n,), device="cuda:0")
order = torch.argsort(key4)
run(rows[order].contiguous(), act[order.cpu().numpy()], "random inside a read")
order = torch.randperm(n, device="cuda:0")
run(rows[order].contiguous(), act[order.cpu().numpy()], "random over all reads")
