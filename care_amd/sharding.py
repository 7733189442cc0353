"""Batch-axis sharding of the captioning path over the GPUs of one node.

Every clip is independent end to end, so the data path needs NO collective: rank r
translates its own contiguous chunk of the global batch with a full weight replica.  The
only exchange is the metrics step: fixed-size per-clip records - token ids, length, score and,
for models with a concept head, the fp32 concept probabilities `preds_attr [B_local, k]` that the
reference's `NoisyOrMIL` criterion scores (misc/Crit/crit_attribute.py:58-89) - are all-gathered
in ONE collective: `torch.distributed` with the `nccl` backend, which is RCCL over xGMI on ROCm
(gloo on CPU in the tests).  Ragged tails are padded to equal per-rank sizes and carried with a
validity column.  A record is (T + 4 + k) int32 = 2.1 KB per clip for k = 500.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int, int]:
    """Contiguous chunk [lo, hi) of rank `rank` and the padded per-rank size."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    hi = min(lo + per, n_items)
    return lo, hi, per


def shard_feats(feats: List[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    lo, hi, _ = shard_bounds(feats[0].shape[0], rank, world)
    return [f[lo:hi] for f in feats]


def pack_records(fed: torch.Tensor, length: torch.Tensor, score: torch.Tensor, per: int,
                 preds_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[n, T+1] int32 tokens, [n] lengths, [n] fp32 scores (+ [n, k] fp32 concept probabilities)
    -> int32 [per, T+4 (+k)] records.

    Columns: T+1 tokens, length, score bits, valid flag, then the k probability bit patterns; rows
    >= n are padding (valid = 0).
    """
    n, width = fed.shape
    k = 0 if preds_attr is None else preds_attr.shape[1]
    rec = torch.zeros(per, width + 3 + k, dtype=torch.int32, device=fed.device)
    rec[:n, :width] = fed
    rec[:n, width] = length.to(torch.int32)
    rec[:n, width + 1] = score.to(torch.float32).contiguous().view(torch.int32)
    rec[:n, width + 2] = 1
    if k:
        rec[:n, width + 3:] = preds_attr.to(torch.float32).contiguous().view(torch.int32)
    return rec


def unpack_records(rec: torch.Tensor, n_attr: int = 0):
    """Inverse of pack_records on the gathered block: (tokens, lengths, scores[, preds_attr])."""
    width = rec.shape[1] - 3 - n_attr
    valid = rec[:, width + 2] == 1
    rec = rec[valid]
    out = (rec[:, :width], rec[:, width], rec[:, width + 1].contiguous().view(torch.float32))
    if n_attr:
        out += (rec[:, width + 3:].contiguous().view(torch.float32),)
    return out


def all_gather_records(rec: torch.Tensor, out: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """All-gather equal-size record blocks; returns the [world * per, cols] concatenation."""
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world = dist.get_world_size()
    if out is None:
        out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    return torch.cat(out, dim=0)


def gather_captions(fed: torch.Tensor, length: torch.Tensor, score: torch.Tensor, n_global: int,
                    preds_attr: Optional[torch.Tensor] = None):
    """Metrics-step exchange: every rank ends with the whole batch's (tokens, lengths, scores) and,
    when `preds_attr` is given, the whole batch's concept probabilities (for concept_metrics)."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    _, _, per = shard_bounds(n_global, 0, world)
    n_attr = 0 if preds_attr is None else preds_attr.shape[1]
    return unpack_records(all_gather_records(pack_records(fed, length, score, per, preds_attr)), n_attr)
