"""Data-parallel tile sharding and detection collation.

The path shards by tile: tiles are independent (per-image PostProcess, build_sam.py:237;
per-image NMS, visualize_prediction.py:150-154), weights are replicated, and the only
exchange is collating detections at the end -- the analogue of the reference's evaluation
gather (inference.py:240-259 via utils/misc.py:180-220).  Here that is ONE fixed-size
all-gather of box records (51 slots x 32 B per tile): no pickle, no size pre-exchange
(all_gather_records, the bench / inference path; gather_detections, the evaluate() merge, sizes
its padded buffer with one 2-word all-reduce first because per-rank image counts differ).
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


EXTRA_WORDS = 4            # per image, behind its 51 x 8 record words: image id (low, high 32 bits), valid flag, reserved


def gather_detections(local: dict) -> dict:
    """Merge per-image detections {image_id: {'boxes' (n,4), 'scores' (n,), 'labels' (n,)}} (n <= 51, PostProcess output)
    across ranks -- the analogue of the reference's evaluation gather (inference.py:240-259 through utils/misc.py:180-220)
    without pickling.  Two collectives: a 2-word all-reduce (MAX) of [images on this rank, error flag], which sizes the
    padded buffer and lets every rank raise TOGETHER when one rank's input is invalid (a rank that raised alone would leave
    the others blocked in the collective), then ONE all-gather of int32 words: per image its 51 x [x0,y0,x1,y1,score,label,
    valid,0] fp32 record (bit pattern) followed by the int64 image id and a valid flag.  Returns the merged dict on every
    rank; an image seen by several ranks (DistributedSampler padding) is kept once."""
    import numpy as np
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1:
        for k, d in local.items():
            if len(d["scores"]) > SLOTS:
                raise ValueError(f"image {k}: {len(d['scores'])} detections exceed the {SLOTS} slots of a record")
        return dict(local)
    ids = sorted(local)
    words = SLOTS * RECORD_FLOATS + EXTRA_WORDS
    rec = torch.zeros((len(ids), SLOTS, RECORD_FLOATS), dtype=torch.float32)
    bad = ""
    for i, k in enumerate(ids):
        d = local[k]
        n = len(d["scores"])
        if n > SLOTS:
            bad = bad or f"image {k}: {n} detections exceed the {SLOTS} slots of a record"
            continue
        rec[i, :n, 0:4] = torch.as_tensor(np.asarray(d["boxes"], dtype=np.float32)).reshape(n, 4)
        rec[i, :n, 4] = torch.as_tensor(np.asarray(d["scores"], dtype=np.float32))
        rec[i, :n, 5] = torch.as_tensor(np.asarray(d["labels"], dtype=np.float32))
        rec[i, :n, 6] = 1.0
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    status = torch.tensor([len(ids), 1 if bad else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(status, op=dist.ReduceOp.MAX)
    cap, any_bad = int(status[0].item()), int(status[1].item())
    if any_bad:                                               # every rank leaves here, none is left inside a collective
        raise ValueError(bad or "gather_detections: another rank holds an image with more detections than a record has slots")
    buf = torch.zeros((cap, words), dtype=torch.int32)
    if ids:
        buf[: len(ids), : SLOTS * RECORD_FLOATS] = rec.view(len(ids), -1).view(torch.int32)
        idt = torch.tensor(ids, dtype=torch.int64)
        buf[: len(ids), SLOTS * RECORD_FLOATS:SLOTS * RECORD_FLOATS + 2] = idt.view(-1, 1).view(torch.int32).view(len(ids), 2)
        buf[: len(ids), SLOTS * RECORD_FLOATS + 2] = 1
    buf = buf.to(dev)
    gathered = torch.empty((world, cap, words), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered.view(-1), buf.view(-1))
    gathered = gathered.cpu()
    merged: dict = {}
    for r in range(world):
        for i in range(cap):
            row = gathered[r, i]
            if int(row[SLOTS * RECORD_FLOATS + 2]) != 1:
                continue
            k = int(row[SLOTS * RECORD_FLOATS:SLOTS * RECORD_FLOATS + 2].contiguous().view(torch.int64).item())
            if k in merged:
                continue
            recs = row[: SLOTS * RECORD_FLOATS].contiguous().view(torch.float32).view(SLOTS, RECORD_FLOATS)
            valid = recs[:, 6] > 0
            merged[k] = {"boxes": recs[valid, 0:4].numpy(), "scores": recs[valid, 4].numpy(),
                         "labels": recs[valid, 5].numpy().astype(np.int64)}
    return merged
