"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm), the
genome sharded by tile-aligned plane ranges, every rank searching all reads on its shard, and ONE
exchange per search: the gather of the hit records to rank 0 over xGMI.

torch is used for the process group and the device buffers that RCCL moves - nothing else.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import TILE_WORDS, HIT_DTYPE, merge_shard_records

RECORD_BYTES = HIT_DTYPE.itemsize


def shard_words(n_words_total, rank, world):
    """Tile-aligned word range [begin, end) of the planes owned by `rank` (same rule as
    PackedGenome.shard_words; windows are owned by the shard that holds their first base)."""
    tiles = (n_words_total + TILE_WORDS - 1) // TILE_WORDS
    b = (tiles * rank // world) * TILE_WORDS
    e = (tiles * (rank + 1) // world) * TILE_WORDS
    return min(b, n_words_total), min(e, n_words_total)


def gather_records(local, group=None, dst=0):
    """local: 1-D uint8 tensor holding this rank's records (device tensor for nccl, CPU for gloo).

    Returns (on dst) a uint8 tensor with the records of rank 0, 1, ... concatenated in rank order and
    the per-rank record counts; (None, counts) elsewhere.  One all_gather of the counts (8 bytes per
    rank) + one grouped send/recv of the variable-length payloads straight into their final place.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([local.numel() // RECORD_BYTES], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        out = torch.empty(sum(counts) * RECORD_BYTES, dtype=torch.uint8, device=local.device)
        offs = np.concatenate([[0], np.cumsum(counts)]) * RECORD_BYTES
        out[offs[dst]:offs[dst + 1]] = local
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group)
               for r in range(world) if r != dst and counts[r] > 0]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, local, dst, group)] if counts[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


def sharded_search(ctx, genome_shard, codes, max_mismatches, extra_pam=None, group=None, device=None,
                   algorithm="auto"):
    """Search all reads on this rank's shard, gather to rank 0, merge there.

    Returns (merged hits on rank 0 | None, local Hits)."""
    hits = genome_shard.search(codes, max_mismatches, extra_pam, algorithm=algorithm)
    n = len(hits)
    local = torch.empty(n * RECORD_BYTES, dtype=torch.uint8, device=device)
    if n:
        hits.copy_to(local.data_ptr(), local.is_cuda)
    gathered, counts = gather_records(local, group)
    merged = None
    if gathered is not None:
        if gathered.is_cuda:
            torch.cuda.current_stream(gathered.device).synchronize()
        merged = merge_shard_records(ctx, gathered.data_ptr(), gathered.is_cuda, counts, len(codes))
    return merged, hits
