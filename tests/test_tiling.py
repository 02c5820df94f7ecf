"""Large-frame front end: host logic on CPU, kernels on the GPU against oracle/tiling_oracle.py."""
import numpy as np
import pytest
import torch

from wildlifemapper_amd import tiling


def test_tile_origins_cover_frame_with_overlap():
    for H, W in [(4000, 6000), (3648, 5472), (1024, 1024), (700, 900), (1025, 2048), (2048, 3000)]:
        org = tiling.tile_origins(H, W, 1024, 128)
        ys = sorted({y for y, _ in org}); xs = sorted({x for _, x in org})
        assert len(org) == len(ys) * len(xs)
        for size, o in ((H, ys), (W, xs)):
            assert o[0] == 0 and o[-1] == max(0, size - 1024)
            assert all(b - a <= 1024 - 128 for a, b in zip(o, o[1:]))          # neighbours overlap by >= 128
            covered = np.zeros(size, bool)
            for a in o:
                covered[a:a + 1024] = True
            assert covered.all()
    assert len(tiling.tile_origins(4000, 6000)) == 35
    with pytest.raises(ValueError):
        tiling.tile_origins(4000, 6000, 1024, 1024)


@pytest.mark.gpu
def test_frame_to_tiles_bit_exact():
    from oracle import tiling_oracle as TO
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, (1500, 2300, 3), dtype=np.uint8)
    org = tiling.tile_origins(1500, 2300) + [(1000, 2000), (-100, -50)]           # also tiles reaching past the frame
    got = tiling.frame_to_tiles(torch.from_numpy(frame).to(dev), torch.tensor(org, dtype=torch.int32)).cpu().numpy()
    want = TO.cut_tiles(frame, org[:-1])
    assert np.array_equal(got[:-1], want)
    assert not got[-1][:, :100, :].any() and not got[-1][:, :, :50].any()
    assert np.array_equal(got[-1][:, 100:, 50:], want[0][:, : 1024 - 100, : 1024 - 50])


@pytest.mark.gpu
def test_merge_tiles_nms_matches_oracle():
    """Synthetic per-tile records: random boxes, some duplicated in the neighbouring tile (same frame box, shifted tile
    coordinates, slightly different score), ties, non-candidates: merged keep list identical to the numpy NMS."""
    from oracle import tiling_oracle as TO
    from wildlifemapper_amd import _native as N
    from wildlifemapper_amd.engine import split_records
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    org = tiling.tile_origins(2048, 3000)
    n = len(org)
    boxes = np.zeros((n, 51, 4), np.float32)
    scores = rng.random((n, 51)).astype(np.float32)
    c = rng.random((n, 51, 2)) * 900 + 50
    wh = rng.random((n, 51, 2)) * 80 + 10
    boxes[..., :2] = c - wh / 2
    boxes[..., 2:] = c + wh / 2
    cand = rng.random((n, 51)) < 0.4
    # duplicates across horizontally neighbouring tiles
    for t in range(n - 1):
        if org[t][0] == org[t + 1][0]:
            dx = org[t + 1][1] - org[t][1]
            for s in range(5):
                boxes[t + 1, s] = boxes[t, s] - np.array([dx, 0, dx, 0], np.float32)
                scores[t + 1, s] = scores[t, s] * (0.99 if s % 2 else 1.0)      # some exact ties
                cand[t, s] = cand[t + 1, s] = True
    rec = torch.zeros((n, 51, 8), dtype=torch.float32)
    rec[..., 0:4] = torch.from_numpy(boxes)
    rec[..., 4] = torch.from_numpy(scores)
    ints = rec.view(torch.int32)
    ints[..., 5] = torch.from_numpy(rng.integers(0, 7, (n, 51)).astype(np.int32))
    ints[..., 6] = torch.from_numpy(np.where(cand, N.FLAG_CONF | N.FLAG_SCORE | N.FLAG_NMS, N.FLAG_CONF).astype(np.int32))
    ints[..., 7] = -1
    merged = tiling.merge_tile_records(rec.to(dev), torch.tensor(org, dtype=torch.int32), 0.4).cpu()
    r = split_records(merged)
    fb, keep = TO.merge(boxes, scores, cand, org, 0.4)
    assert np.array_equal(r["boxes"].reshape(-1, 4).numpy(), fb)
    flags, rank = r["flags"].reshape(-1), r["nms_rank"].reshape(-1)
    got = torch.nonzero((flags & N.FLAG_MERGED) != 0).flatten()
    got = got[torch.argsort(rank[got])].numpy()
    assert np.array_equal(got, keep)
    assert len(keep) < int(cand.sum())                                           # the duplicates were merged away
    assert int((rank >= 0).sum()) == len(keep)


@pytest.mark.gpu
def test_detect_frame_equals_per_tile_path_plus_oracle_merge():
    """End to end on a 2048 x 3000 synthetic frame (ViT-B): detect_frame = cut tiles (oracle) -> model.detect per tile ->
    oracle merge."""
    from oracle import tiling_oracle as TO
    from wildlifemapper_amd import synth, _native as N
    from wildlifemapper_amd.engine import split_records
    from wildlifemapper_amd.segment_anything import sam_model_registry
    from wildlifemapper_amd.segment_anything.network import MedSAM
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_b").items()}
    sam, _, _ = sam_model_registry["vit_b"](None, None)
    m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
    m.load_state_dict(sd, strict=True)
    m._hub.set_precision("fp16")
    rng = np.random.default_rng(9)
    frame = rng.integers(0, 256, (2048, 3000, 3), dtype=np.uint8)
    out = tiling.detect_frame(m, torch.from_numpy(frame).to(dev), overlap=128, batch=4)
    org = tiling.tile_origins(2048, 3000)
    assert out["origins"].cpu().tolist() == [list(o) for o in org] and len(org) == 12
    tiles = torch.from_numpy(TO.cut_tiles(frame, org)).to(dev)
    rec = torch.cat([m.detect(tiles[i:i + 4])["records"] for i in range(0, len(org), 4)]).cpu()
    r = split_records(rec)
    fb, keep = TO.merge(r["boxes"].numpy(), r["scores"].numpy(), ((r["flags"] & N.FLAG_NMS) != 0).numpy(), org, 0.4)
    assert len(keep) > 0
    np.testing.assert_array_equal(out["boxes"].cpu().numpy(), fb[keep])
    np.testing.assert_array_equal(out["tile"].cpu().numpy(), keep // 51)
    m._hub.close()
