"""Data-parallel scoring across the GPUs of one node: one process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI on ROCm), utterances sharded by
index, weights replicated, and ONE small collective to put the scores back together.

The reference has no live counterpart: its DDP eval keeps per-rank metrics
(trainer.py:85-132) and the gather it intended is commented out (trainer.py:119-120,
ddp_util.py).  Unlike its DistributedSampler (main.py:33-39) nothing is duplicated or
shuffled: ranks take i = r (mod W), short shards are padded with the sentinel -1, and
the merged list is restored to dataset order before it is written.

Hardware queues: ROCm maps a process's HIP streams onto 4 hardware queues in order of first use and shares them from the fifth
stream on.  RCCL brings up streams of its own; a stream first used after that can share the trunk stream's queue: results stay
identical but whatever was meant to run beside the trunk runs behind it (measured: the head / trunk overlap of
Engine.forward_overlapped lost whole, 4.95 -> 5.20 ms per student step).  So: ``afx.engine.side_stream(device)`` BEFORE
``init_process_group`` (bench.py does).  Neither GPU_MAX_HW_QUEUES=8 nor high-priority streams: more than 4 queues in use
made the two-stream step 2x slower (tools/diag_dist_overlap.py, profiles/r03_k_dist_overlap_hw_queues.txt).
"""
import torch
import torch.distributed as dist

SENTINEL = -1


def shard_indices(n_items, rank, world):
    """Indices rank, rank+W, ... padded with -1 so every rank holds ceil(n/W) slots."""
    per = (n_items + world - 1) // world
    idx = torch.arange(rank, rank + per * world, world, dtype=torch.int64)
    idx[idx >= n_items] = SENTINEL
    return idx


def all_gather_scores(idx, scores, world=None, group=None):
    """(idx:int32[n], score:fp32[n]) on every rank -> the W*n pairs of all ranks, rank-
    major.  One collective: both halves travel bit-cast in a single int32 buffer
    (8 bytes per utterance -- latency-bound; bandwidth of the xGMI links is irrelevant)."""
    world = world or dist.get_world_size(group)
    n = idx.numel()
    payload = torch.cat([idx.to(torch.int32).reshape(-1), scores.to(torch.float32).contiguous().reshape(-1).view(torch.int32)])
    out = torch.empty(world * 2 * n, dtype=torch.int32, device=payload.device)
    dist.all_gather_into_tensor(out, payload, group=group)
    out = out.view(world, 2, n)
    return out[:, 0].reshape(-1), out[:, 1].contiguous().reshape(-1).view(torch.float32)


def merge_scores(idx, scores):
    """Drop sentinels and restore dataset order (shuffle=False, main.py:200-201)."""
    keep = idx >= 0
    idx, scores = idx[keep], scores[keep]
    order = torch.argsort(idx)
    return idx[order], scores[order]
