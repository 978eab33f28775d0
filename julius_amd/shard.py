"""Batch-of-utterances sharding across the GPUs of one node.

Utterances are independent, so the only multi-GPU structure is a partition of
the batch: rank r decodes utterances r, r+W, r+2W, ... on its own GPU with its
own replica of the model and lexicon; there is NO data-path collective and no
cross-GPU trellis.  torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests) is used only to gather fixed-size result
records and counters at the end.
"""
from __future__ import annotations

import numpy as np

MAXSEQ = 150  # MAXSEQNUM, libsent/include/sent/speech.h:50
REC = 4 + MAXSEQ  # status, wnum, frames, score bits, wseq[150]


def shard_indices(nutt: int, rank: int, world: int) -> np.ndarray:
    """Utterance ids decoded by `rank` (round robin: balances long/short inputs)."""
    return np.arange(rank, nutt, world, dtype=np.int64)


def pack_results(results) -> np.ndarray:
    """jamd_pass1_result-like objects -> int32 [n][REC] records."""
    out = np.zeros((len(results), REC), np.int32)
    for i, r in enumerate(results):
        out[i, 0], out[i, 1], out[i, 2] = r.status, r.wnum, r.frames
        out[i, 3] = np.float32(r.score).view(np.int32)
        out[i, 4:4 + r.wnum] = np.asarray(r.wseq[:r.wnum], np.int32)
    return out


def unpack_results(rec: np.ndarray):
    return [dict(status=int(r[0]), wnum=int(r[1]), frames=int(r[2]), score=float(r[3:4].view(np.float32)[0]),
                 wseq=r[4:4 + int(r[1])].copy()) for r in np.asarray(rec, np.int32)]


def gather_results(local_rec: np.ndarray, nutt: int, rank: int, world: int, device="cpu"):
    """All ranks -> every rank gets the [nutt][REC] table in utterance order.
    Shards differ in length by at most one, so records are padded to the longest
    shard and all_gather'ed once."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return np.asarray(local_rec, np.int32)
    per = (nutt + world - 1) // world
    buf = torch.zeros((per, REC), dtype=torch.int32, device=device)
    if len(local_rec):
        buf[:len(local_rec)] = torch.from_numpy(np.ascontiguousarray(local_rec, np.int32)).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    table = np.zeros((nutt, REC), np.int32)
    for r in range(world):
        idx = shard_indices(nutt, r, world)
        table[idx] = parts[r][:len(idx)].cpu().numpy()
    return table


def reduce_counters(frames: int, tokens: int, device="cpu"):
    """Whole-node frame / token counters (all_reduce sum)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([frames, tokens], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0]), int(t[1])
