#!/bin/bash
mkdir -p gpurun_out/r2x
timeout -k 10 300 python - > gpurun_out/r2x/bw.log 2>&1 <<'PY'
import torch, time
def bench(f, n=10):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
N = 8 << 30
a = torch.empty(N, dtype=torch.uint8, device="cuda"); b = torch.empty(N, dtype=torch.uint8, device="cuda")
ai = a.view(torch.int64); bi = b.view(torch.int64)
ms = bench(lambda: ai.fill_(7)); print("fill 8 GiB", round(ms, 3), "ms", round(N / ms / 1e9, 2), "TB/s")
ms = bench(lambda: bi.copy_(ai)); print("copy 8 GiB", round(ms, 3), "ms", round(2 * N / ms / 1e9, 2), "TB/s (r+w)")
ms = bench(lambda: ai.sum()); print("read 8 GiB", round(ms, 3), "ms", round(N / ms / 1e9, 2), "TB/s")
c = torch.empty(N // 4, dtype=torch.uint8, device="cuda").view(torch.int64)
ms = bench(lambda: torch.add(ai[: N // 32], 1, out=c[: N // 32])); print("r 2 + w 2 GiB", round(ms, 3))
PY
cat gpurun_out/r2x/bw.log
