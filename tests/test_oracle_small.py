"""Pin the CPU oracle against fixtures produced by the reference's own modules.

tests/golden/small_ops.npz was written by oracle/gen_golden.py, which ran the
reference `modeling` classes (Block, window_partition, add_decomposed_rel_pos,
MLPBlock, LayerNorm2d, TwoWayTransformer, MLP, PromptEncoder) on seeded inputs.
"""
import os

import numpy as np
import pytest
import torch

from oracle import wm_oracle as O
from wildlifemapper_amd import synth


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "small_ops.npz"))


def _weights(prefix, names_shapes):
    return {n: torch.from_numpy(synth.make_weight(prefix + n, s)) for n, s in names_shapes.items()}


def _block_weights(tag, dim, heads, size):
    hd = dim // heads
    p = "image_encoder.blocks.0."
    shapes = {
        p + "norm1.weight": (dim,), p + "norm1.bias": (dim,),
        p + "attn.rel_pos_h": (2 * size - 1, hd), p + "attn.rel_pos_w": (2 * size - 1, hd),
        p + "attn.qkv.weight": (3 * dim, dim), p + "attn.qkv.bias": (3 * dim,),
        p + "attn.proj.weight": (dim, dim), p + "attn.proj.bias": (dim,),
        p + "norm2.weight": (dim,), p + "norm2.bias": (dim,),
        p + "mlp.lin1.weight": (4 * dim, dim), p + "mlp.lin1.bias": (4 * dim,),
        p + "mlp.lin2.weight": (dim, 4 * dim), p + "mlp.lin2.bias": (dim,),
    }
    return _weights(f"small.{tag}.", shapes)


@pytest.mark.parametrize("tag,grid,ws", [("win", 20, 14), ("glob", 12, 0)])
def test_encoder_block(fx, tag, grid, ws):
    W = _block_weights(tag, 64, 2, ws if ws else grid)
    cfg = O.OracleCfg(embed_dim=64, depth=1, num_heads=2, global_attn_indexes=() if ws else (0,), grid=grid, window=14)
    y = O.encoder_block(torch.from_numpy(fx[f"block_{tag}_x"]), W, 0, cfg)
    np.testing.assert_allclose(y.numpy(), fx[f"block_{tag}_y"], rtol=2e-5, atol=2e-5)


def test_window_roundtrip(fx):
    x = torch.from_numpy(fx["winpart_x"])
    w, n = O.to_windows(x, 4)
    np.testing.assert_array_equal(w.numpy(), fx["winpart_w"])
    back = O.from_windows(w, 4, n, 9)
    np.testing.assert_array_equal(back.numpy(), fx["winpart_back"])
    np.testing.assert_array_equal(back.numpy(), fx["winpart_x"])


def test_rel_pos(fx):
    q = torch.from_numpy(fx["relpos_q"]).reshape(3, 5, 5, 8)
    Rh = O.rel_pos_table(5, torch.from_numpy(fx["relpos_rh"]))
    Rw = O.rel_pos_table(5, torch.from_numpy(fx["relpos_rw"]))
    rel_h = torch.einsum("bhwc,hkc->bhwk", q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", q, Rw)
    a = torch.from_numpy(fx["relpos_attn"]).view(3, 5, 5, 5, 5) + rel_h[..., :, None] + rel_w[..., None, :]
    np.testing.assert_allclose(a.reshape(3, 25, 25).numpy(), fx["relpos_out"], rtol=1e-5, atol=1e-5)


def test_mlp_gelu(fx):
    W = _weights("small.mlp.", {"lin1.weight": (64, 16), "lin1.bias": (64,), "lin2.weight": (16, 64), "lin2.bias": (16,)})
    cfg = O.OracleCfg()
    x = torch.from_numpy(fx["mlp_x"])
    y = O.linear(O.gelu_erf(O.linear(x, W["lin1.weight"], W["lin1.bias"], cfg)), W["lin2.weight"], W["lin2.bias"], cfg)
    np.testing.assert_allclose(y.numpy(), fx["mlp_y"], rtol=1e-5, atol=1e-6)


def test_layernorm2d(fx):
    W = _weights("small.ln2d.norm.", {"weight": (8,), "bias": (8,)})
    x = torch.from_numpy(fx["ln2d_x"])
    y = O.layer_norm(x.permute(0, 2, 3, 1), W["weight"], W["bias"], 1e-6).permute(0, 3, 1, 2)
    np.testing.assert_allclose(y.numpy(), fx["ln2d_y"], rtol=1e-5, atol=1e-6)


def _dec_weights(E, internal_half, mlp):
    shapes = {}
    t = "mask_decoder.transformer."

    def attn(prefix, internal):
        for p in ("q_proj", "k_proj", "v_proj"):
            shapes[prefix + p + ".weight"] = (internal, E)
            shapes[prefix + p + ".bias"] = (internal,)
        shapes[prefix + "out_proj.weight"] = (E, internal)
        shapes[prefix + "out_proj.bias"] = (E,)

    for i in range(2):
        L = f"{t}layers.{i}."
        attn(L + "self_attn.", E)
        attn(L + "cross_attn_token_to_image.", internal_half)
        attn(L + "cross_attn_image_to_token.", internal_half)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            shapes[L + n + ".weight"] = (E,)
            shapes[L + n + ".bias"] = (E,)
        shapes[L + "mlp.lin1.weight"] = (mlp, E)
        shapes[L + "mlp.lin1.bias"] = (mlp,)
        shapes[L + "mlp.lin2.weight"] = (E, mlp)
        shapes[L + "mlp.lin2.bias"] = (E,)
    attn(t + "final_attn_token_to_image.", internal_half)
    shapes[t + "norm_final_attn.weight"] = (E,)
    shapes[t + "norm_final_attn.bias"] = (E,)
    dims = [E, E, E, 8]
    for j in range(3):
        shapes[f"mask_decoder.class_embed.layers.{j}.weight"] = (dims[j + 1], dims[j])
        shapes[f"mask_decoder.class_embed.layers.{j}.bias"] = (dims[j + 1],)
    return _weights("small.dec.", shapes)


def test_two_way_transformer_and_head(fx):
    W = _dec_weights(32, 16, 64)
    cfg = O.OracleCfg(dec_heads=4, dec_depth=2, grid=6)
    q, k = O.two_way_transformer(torch.from_numpy(fx["tw_src"]), torch.from_numpy(fx["tw_pos"]),
                                 torch.from_numpy(fx["tw_tok"]), W, cfg)
    np.testing.assert_allclose(q.numpy(), fx["tw_queries"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(k.numpy(), fx["tw_keys"], rtol=2e-5, atol=2e-5)
    y = O.mlp_head(q, W, "mask_decoder.class_embed.", cfg)
    np.testing.assert_allclose(y.numpy(), fx["head_y"], rtol=2e-5, atol=2e-4)


def test_dense_pe(fx):
    g = torch.from_numpy(synth.make_weight("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", (2, 128)))
    pe = O.dense_pe(g, 64)
    assert pe.shape == (1, 256, 64, 64)
    f = pe.reshape(-1)
    step = max(1, f.numel() // 8192)
    np.testing.assert_allclose(f[::step][:8192].numpy(), fx["dense_pe_sample"], rtol=0, atol=2e-5)


def test_oracle_end_to_end_vs_reference_fixture(golden_dir):
    """The oracle's whole path (fft -> encoder -> decoder) on ViT-B tile 0 against tests/golden/e2e_vit_b.npz, which
    oracle/gen_golden.py produced by running the REFERENCE's own ImageEncoderViT / MaskDecoder / PromptEncoder here:
    logits, boxes, the stem, the HFC adaptor's output, every block tap and the embedding.  This is the end-to-end pin
    DESIGN.md section 2 cites (fp32 on both sides: only summation order differs)."""
    fx = np.load(os.path.join(golden_dir, "e2e_vit_b.npz"))
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_b").items()}
    x = torch.from_numpy(synth.make_batch(int(fx["first_tile"]), 1))
    cfg = O.OracleCfg.from_model_type("vit_b")
    taps = {}
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    out = O.model_forward(x, sd, cfg, taps)

    def sample(t, n):
        f = t.detach().reshape(-1)
        step = max(1, f.numel() // n)
        return f[::step][:n].float().numpy()

    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    # the fixture holds 2 tiles; strided samples of a 2-tile tensor are not samples of its first tile, so the sampled keys
    # are compared through a 2-tile run only where cheap: here tile 0 is checked on the un-sampled outputs ...
    assert rel(out["pred_logits"].numpy(), fx["pred_logits"][:1]) < 5e-6
    assert np.abs(out["pred_boxes"].numpy() - fx["pred_boxes"][:1]).max() < 5e-6
    np.testing.assert_allclose(taps["embedding"].double().mean(dim=(2, 3)).numpy(), fx["emb_chan_mean"][:1], atol=2e-6)
    # ... and on the sampled keys whose sampling stride keeps tile 0's values in the first half of the sample
    n = int(fx["n_tiles"])
    for key, t, cnt in (("hfc_sample", taps["hfc"], 8192), ("stem_sample", taps["stem"], 8192), ("emb_sample", taps["embedding"], 16384),
                        ("block0_sample", taps["block0"], 2048), ("block5_sample", taps["block5"], 2048), ("block11_sample", taps["block11"], 2048)):
        step2 = max(1, t.numel() * n // cnt)                 # stride the generator used on the n-tile tensor
        mine = t.detach().reshape(-1)[::step2].float().numpy()
        ref = fx[key][: len(mine)]
        assert rel(mine, ref) < 2e-5, (key, rel(mine, ref))
    det = O.detect(O.postprocess(out["pred_logits"], out["pred_boxes"], torch.tensor([[1024, 1024]]))[0])
    assert det["nms_index"].tolist() == fx["pp0_nms_index"].tolist()


def test_oracle_end_to_end_matches_reference_fixture_vit_l(golden_dir):
    """The oracle on the registry's third entry (ViT-L: 1024 wide, 24 blocks, hd 64) against tests/golden/e2e_vit_l.npz,
    which oracle/gen_golden.py --only vit_l produced by running the reference's modules on tile 5: logits, boxes, per-channel
    embedding means, NMS list (fp32 on both sides)."""
    fx = np.load(os.path.join(golden_dir, "e2e_vit_l.npz"))
    assert int(fx["n_tiles"]) == 1
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_l").items()}
    x = torch.from_numpy(synth.make_batch(int(fx["first_tile"]), 1))
    taps = {}
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    out = O.model_forward(x, sd, O.OracleCfg.from_model_type("vit_l"), taps)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel(out["pred_logits"].numpy(), fx["pred_logits"]) < 5e-6
    assert np.abs(out["pred_boxes"].numpy() - fx["pred_boxes"]).max() < 5e-6
    np.testing.assert_allclose(taps["embedding"].double().mean(dim=(2, 3)).numpy(), fx["emb_chan_mean"], atol=2e-6)
    det = O.detect(O.postprocess(out["pred_logits"], out["pred_boxes"], torch.tensor([[1024, 1024]]))[0])
    assert det["nms_index"].tolist() == fx["pp0_nms_index"].tolist()


def test_pil_resize_restatement(golden_dir):
    """oracle/pil_resize.py (the val transform's resize, dataloader_coco.py:288 -> PIL bilinear) against vectors PIL itself
    produced (tests/golden/resize_pil.npz, oracle/gen_golden.py --only resize): bit-exact, and the output-size rule of
    augmentation.py:80-99 on the dataset's frame sizes."""
    from oracle import pil_resize as R
    fx = np.load(os.path.join(golden_dir, "resize_pil.npz"))
    for i, (h, w, size, mx) in enumerate(fx["cases"]):
        img, want = fx[f"in{i}"], fx[f"out{i}"]
        assert img.shape == (h, w, 3)
        oh, ow = R.get_size_with_aspect_ratio((int(w), int(h)), int(size), int(mx))
        assert want.shape == (oh, ow, 3)
        assert np.array_equal(R.resize_bilinear_u8(img, oh, ow), want), i
    assert R.get_size_with_aspect_ratio((5472, 3648), 768, 768) == (512, 768)      # coco_annotations/val.json frames
    assert R.get_size_with_aspect_ratio((6000, 4000), 768, 768) == (512, 768)
    assert R.get_size_with_aspect_ratio((3648, 5472), 768, 768) == (768, 512)
    assert R.get_size_with_aspect_ratio((1000, 1000), 768, 768) == (768, 768)


def test_postprocess_restatement_vs_reference_fixture(golden_dir):
    """A19 pinned (round 3): oracle.postprocess against tests/golden/postprocess_ref.npz, which oracle/gen_golden.py
    --only postprocess produced by running the reference's own PostProcess class (build_sam.py:212-258) and
    box_cxcywh_to_xyxy (utils/box_ops.py:9-13), taken from the reference files by definition node, on the logits / boxes of
    the end-to-end fixtures, on random heads with non-square target sizes and on a tile with nothing above the confidence
    threshold.  Bit-exact: same torch ops in the same order."""
    fx = np.load(os.path.join(golden_dir, "postprocess_ref.npz"))
    assert int(fx["pinned"]) == 1
    for tag in fx["cases"]:
        tag = str(tag)
        lg, bx, ts = (torch.from_numpy(fx[f"{tag}_{k}"]) for k in ("logits", "boxes", "sizes"))
        got = O.postprocess(lg, bx, ts)
        for b, r in enumerate(got):
            assert np.array_equal(r["scores"].numpy(), fx[f"{tag}_pp{b}_scores"]), (tag, b)
            assert np.array_equal(r["labels"].numpy(), fx[f"{tag}_pp{b}_labels"]), (tag, b)
            assert np.array_equal(r["boxes"].numpy().reshape(-1, 4), fx[f"{tag}_pp{b}_boxes"]), (tag, b)
    assert len(fx["empty_pp0_scores"]) == 0 and fx["empty_pp0_boxes"].shape == (0, 4)
    x = torch.from_numpy(fx["cxcywh_in"])
    cx, cy, w, h = x.unbind(-1)
    assert np.array_equal(torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1).numpy(), fx["cxcywh_out"])
    # the pp*_ keys inside the end-to-end fixtures (written by the oracle, pinned=0 there) equal the reference's outputs
    e = np.load(os.path.join(golden_dir, "e2e_vit_h.npz"))
    assert np.array_equal(e["pp0_scores"], fx["e2e_vit_h_pp0_scores"]) and np.array_equal(e["pp0_boxes"], fx["e2e_vit_h_pp0_boxes"])
