"""Data-parallel tile sharding and detection collation.

The path shards by tile: tiles are independent (per-image PostProcess, build_sam.py:237;
per-image NMS, visualize_prediction.py:150-154), weights are replicated, and the only
exchange is collating detections at the end -- the analogue of the reference's evaluation
gather (inference.py:240-259 via utils/misc.py:180-220).  Here that is ONE fixed-size
all-gather of box records (51 slots x 32 B per tile): no pickle, no size pre-exchange.
One process per GPU; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist

RECORD_FLOATS = 8          # sizeof(wm_box_record) / 4
SLOTS = 51


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the process group if world > 1."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_tiles: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of tiles owned by `rank` (first ranks take the remainder)."""
    q, r = divmod(n_tiles, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def max_shard(n_tiles: int, world: int) -> int:
    return (n_tiles + world - 1) // world


def all_gather_records(records: torch.Tensor, n_tiles: int, rank: int, world: int) -> torch.Tensor:
    """records: this rank's (n_local, 51, 8) float32 raw records.  Returns (n_tiles, 51, 8) in global
    tile order on every rank.  Shards are padded to the largest shard so the collective is fixed-size."""
    if world == 1:
        return records
    cap = max_shard(n_tiles, world)
    # RCCL gathers device buffers directly; the gloo backend (CPU tests, single-GPU rehearsals) needs host memory
    home = records.device
    work = torch.device("cpu") if (dist.get_backend() == "gloo" and records.is_cuda) else home
    padded = torch.zeros((cap, SLOTS, RECORD_FLOATS), dtype=torch.float32, device=work)
    padded[: records.shape[0]] = records.to(work)
    out = torch.empty((world, cap, SLOTS, RECORD_FLOATS), dtype=torch.float32, device=work)
    dist.all_gather_into_tensor(out.view(-1), padded.view(-1))
    out = out.to(home)
    parts: List[torch.Tensor] = []
    for r in range(world):
        s, e = shard_range(n_tiles, r, world)
        parts.append(out[r, : e - s])
    return torch.cat(parts, dim=0)


def gather_detections(local: dict) -> dict:
    """Merge per-image detections {image_id: {'boxes' (n,4), 'scores' (n,), 'labels' (n,)}} (n <= 51, PostProcess output)
    across ranks: fixed-size records (51 slots x [x0,y0,x1,y1,score,label,valid,0] fp32 per image) + int64 image ids, padded
    to the largest per-rank image count, ONE all-gather each -- the analogue of the reference's evaluation gather
    (inference.py:240-259 through utils/misc.py:180-220) without pickling.  Returns the merged dict on every rank; an image
    seen by several ranks (DistributedSampler padding) is kept once."""
    import numpy as np
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        return dict(local)
    ids = sorted(local)
    rec = torch.zeros((len(ids), SLOTS, RECORD_FLOATS), dtype=torch.float32)
    for i, k in enumerate(ids):
        d = local[k]
        n = len(d["scores"])
        if n > SLOTS:
            raise ValueError(f"image {k}: {n} detections exceed the {SLOTS} slots of a record")
        rec[i, :n, 0:4] = torch.as_tensor(np.asarray(d["boxes"], dtype=np.float32)).reshape(n, 4)
        rec[i, :n, 4] = torch.as_tensor(np.asarray(d["scores"], dtype=np.float32))
        rec[i, :n, 5] = torch.as_tensor(np.asarray(d["labels"], dtype=np.float32))
        rec[i, :n, 6] = 1.0
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    count = torch.tensor([len(ids)], dtype=torch.int64, device=dev)
    cap_t = count.clone()
    dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
    cap = int(cap_t.item())
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, count)
    pad_rec = torch.zeros((cap, SLOTS, RECORD_FLOATS), dtype=torch.float32, device=dev)
    pad_ids = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    if ids:
        pad_rec[: len(ids)] = rec.to(dev)
        pad_ids[: len(ids)] = torch.tensor(ids, dtype=torch.int64, device=dev)
    all_rec = torch.empty((world, cap, SLOTS, RECORD_FLOATS), dtype=torch.float32, device=dev)
    all_ids = torch.empty((world, cap), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_rec.view(-1), pad_rec.view(-1))
    dist.all_gather_into_tensor(all_ids.view(-1), pad_ids.view(-1))
    all_rec, all_ids, counts = all_rec.cpu(), all_ids.cpu(), counts.cpu()
    merged: dict = {}
    for r in range(world):
        for i in range(int(counts[r])):
            k = int(all_ids[r, i])
            if k in merged:
                continue
            valid = all_rec[r, i, :, 6] > 0
            merged[k] = {"boxes": all_rec[r, i, valid, 0:4].numpy(), "scores": all_rec[r, i, valid, 4].numpy(),
                         "labels": all_rec[r, i, valid, 5].numpy().astype(np.int64)}
    return merged
