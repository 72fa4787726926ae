"""Vocab-row sharding across the GPUs of one node (one process per GPU).

Target-vocab rows are independent in every configuration the kept API supports, so
the path shards by rows: rank p computes rows [p*ceil(N/P), (p+1)*ceil(N/P)) with
replicated weights and source embeddings, and ONE all-gather per output matrix
(RCCL over xGMI; backend "nccl" is RCCL on ROCm) reassembles the full [N, E]
matrices on every GPU.  The reference shards the row batch over local devices the
same way (scripts/transfer.py:90-91, zett/utils.py:26) and pads the last shard
(scripts/transfer.py:63-67).  Per-row results do not depend on the shard a row
lands in, so the gathered matrix is bit-identical for every P.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    per = -(-int(n_rows) // int(world))
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def all_gather_rows(local: torch.Tensor, n_rows: int, per: int, group=None) -> torch.Tensor:
    """All-gather row shards of equal nominal height `per`; returns the first n_rows rows."""
    world = dist.get_world_size(group)
    if local.shape[0] != per:                      # last shard(s): pad to the nominal height
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local, group=group)
    return full[:n_rows]


def predict_sharded(predict, target_surface_forms: torch.Tensor, group=None):
    """Run `predict(rows) -> (pred_in, pred_out | None, bias)` on this rank's row shard
    and all-gather the full result on every rank.

    `predict` is typically ``lambda rows: hypernet(rows, source_embeddings=..., lang_index=...)``.
    Without an initialised process group this is just ``predict(target_surface_forms)``.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return predict(target_surface_forms)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = target_surface_forms.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    per = shard_bounds(n, world, 0)[1]
    rows = target_surface_forms[lo:hi]
    if hi - lo == 0:                               # more ranks than rows: compute one dummy row
        rows = target_surface_forms[:1]
    pred_in, pred_out, bias = predict(rows)
    if hi - lo == 0:
        pred_in, bias = pred_in[:0], bias[:0]
        pred_out = None if pred_out is None else pred_out[:0]
    full_in = all_gather_rows(pred_in, n, per, group)
    full_out: Optional[torch.Tensor] = None if pred_out is None else all_gather_rows(pred_out, n, per, group)
    full_bias = all_gather_rows(bias, n, per, group)
    return full_in, full_out, full_bias
