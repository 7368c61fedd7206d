"""Patch-index sharding + the per-round all-gather of refined-patch records.

The refinement path has no data dependence between patches (reference
src/hpmvs/PatchOptimizer.cpp:78-103 reads only its own Patch3d and the immutable scene; the
reference itself runs it as an OpenMP loop over seeds, src/hpmvs/Scene.cpp:114).  One process per
GPU refines a contiguous block of the batch against its own replica of the scene; the only
exchange is ONE all-gather per expansion round of fixed 128-byte records, so that every rank holds
the full refined set before the host scheduler inserts it (RCCL over xGMI on the GPU box, gloo in
the CPU tests).  No all-reduce, no ring dependency.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 128
MAX_RECORD_IMAGES = 32  # ids stored as uint16


def shard_range(n: int, rank: int, world: int):
    """Contiguous block partition [lo, hi) of n patches."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(center, normal, color, scale, fmin, ok, n_images, images) -> torch.Tensor:
    """SoA outputs -> [n, 128] uint8 records (same layout on every rank, any device)."""
    n = scale.shape[0]
    rec = torch.zeros((n, RECORD_BYTES), dtype=torch.uint8, device=scale.device)
    rec[:, 0:16] = center.contiguous().view(torch.uint8).view(n, 16)
    rec[:, 16:32] = normal.contiguous().view(torch.uint8).view(n, 16)
    rec[:, 32:44] = color.contiguous().view(torch.uint8).view(n, 12)
    rec[:, 44:48] = scale.contiguous().view(torch.uint8).view(n, 4)
    rec[:, 48:56] = fmin.contiguous().view(torch.uint8).view(n, 8)
    rec[:, 56] = ok.to(torch.uint8)
    rec[:, 57] = n_images.to(torch.uint8)
    img16 = images[:, :MAX_RECORD_IMAGES].to(torch.int16).contiguous()
    rec[:, 64:64 + 2 * img16.shape[1]] = img16.view(torch.uint8).view(n, -1)
    return rec


def unpack_records(rec: torch.Tensor) -> dict:
    rec = rec.cpu().contiguous()
    n = rec.shape[0]
    f32 = lambda a, b, c: rec[:, a:b].contiguous().view(torch.float32).view(n, c).numpy()
    return dict(center=f32(0, 16, 4), normal=f32(16, 32, 4), color=f32(32, 44, 3), scale=f32(44, 48, 1)[:, 0],
                fmin=rec[:, 48:56].contiguous().view(torch.float64).view(n).numpy(),
                ok=rec[:, 56].numpy().astype(bool), n_images=rec[:, 57].numpy().astype(np.int32),
                images=rec[:, 64:128].contiguous().view(torch.int16).view(n, MAX_RECORD_IMAGES).numpy())


def all_gather_records(local: torch.Tensor, counts=None) -> torch.Tensor:
    """One collective per round.  Equal shard sizes use all_gather_into_tensor; ragged shards
    (n not divisible by the world size) are padded to the largest shard and trimmed."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * local.shape[0], RECORD_BYTES), dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    m = max(counts)
    padded = torch.zeros((m, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * m, RECORD_BYTES), dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m: r * m + counts[r]] for r in range(world)], dim=0)
