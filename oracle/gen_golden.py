#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules here.

Runs only in the build container (needs /root/reference, which does not exist
on the GPU box).  The reference `modeling` sub-package is imported with the
recipe of SURVEY.md §8c (bypassing segment_anything/__init__.py, which needs
torchvision); build-owned synthetic weights (wildlifemapper_amd/synth.py) are
loaded into the reference modules with load_state_dict, the modules are run on
build-owned synthetic tiles, and inputs + expected outputs are stored as small
.npz fixtures.  Nothing from the reference's source text is stored.

What the reference cannot produce here (torchvision missing): MedSAM.fft's
Grayscale and the NMS step.  For those the fixture stores the oracle's own
output, flagged `pinned=0`.

Round 3: `--only postprocess` pins A19.  The reference's PostProcess class (build_sam.py:212-258) and
box_cxcywh_to_xyxy (utils/box_ops.py:9-13) are torch-only bodies inside modules that import torchvision at the top,
so they are taken out of the reference's files with `ast` (the class / function definition nodes, compiled and
executed here, nothing stored) and run on the logits / boxes the end-to-end fixtures already hold plus edge cases;
outputs go to postprocess_ref.npz with pinned=1.  NMS (torchvision.ops.nms) and Grayscale stay unpinned.

Usage:  python oracle/gen_golden.py [--only small|postprocess|vit_b|vit_l|vit_h|vit_h_tiles|vit_h_seed1|vit_h_smooth|vit_h_padded|vit_h_outlier] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/wildlifemapper/segment_anything"

from wildlifemapper_amd import synth  # noqa: E402
from oracle import wm_oracle as O     # noqa: E402


def ref_modeling():
    sys.path.insert(0, REF)
    import modeling  # type: ignore  # the reference's package, imported in place
    return modeling


def sample(t: torch.Tensor, n: int = 4096) -> np.ndarray:
    """Deterministic strided sample of a tensor (flattened), at most n values."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float32).numpy().copy()


def stats(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).mean().sqrt().item(),
                     t.min().item(), t.max().item()], dtype=np.float64)


def load_synth(module: torch.nn.Module, prefix: str, seed: int = 0, profile: str = "baseline") -> dict:
    """Fill every tensor of module.state_dict() with synth.make_weight(prefix+name)."""
    sd = {}
    for k, v in module.state_dict().items():
        sd[k] = torch.from_numpy(synth.make_weight(prefix + k, tuple(v.shape), seed, profile))
    module.load_state_dict(sd, strict=True)
    return {prefix + k: v for k, v in sd.items()}


# ----------------------------------------------------------------------------
def gen_small(out: str) -> None:
    """Per-op fixtures at reduced width (SURVEY.md §8c 'Golden vectors to commit')."""
    M = ref_modeling()
    from modeling.image_encoder import Block, window_partition, window_unpartition, add_decomposed_rel_pos
    from modeling.common import MLPBlock, LayerNorm2d
    from modeling.box_decoder import MLP
    g = torch.Generator().manual_seed(7)
    fx = {}

    # --- encoder Block, windowed with padding (grid 20 -> padded 28), and global (grid 12)
    dim, heads = 64, 2
    for tag, grid, ws in (("win", 20, 14), ("glob", 12, 0)):
        blk = Block(dim=dim, num_heads=heads, mlp_ratio=4.0, qkv_bias=True,
                    norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), act_layer=torch.nn.GELU,
                    use_rel_pos=True, rel_pos_zero_init=True, window_size=ws,
                    input_size=(grid, grid)).eval()
        load_synth(blk, f"small.{tag}.image_encoder.blocks.0.")
        x = torch.randn(2, grid, grid, dim, generator=g)
        with torch.no_grad():
            y = blk(x)
        fx[f"block_{tag}_x"] = x.numpy()
        fx[f"block_{tag}_y"] = y.numpy()

    # --- window partition / unpartition round trip on an odd size
    x = torch.randn(1, 9, 9, 3, generator=g)
    w, pad_hw = window_partition(x, 4)
    fx["winpart_x"] = x.numpy()
    fx["winpart_w"] = w.numpy()
    fx["winpart_back"] = window_unpartition(w, 4, pad_hw, (9, 9)).numpy()

    # --- decomposed rel-pos on its own
    q = torch.randn(3, 5 * 5, 8, generator=g)
    attn = torch.randn(3, 25, 25, generator=g)
    rh = torch.randn(9, 8, generator=g)
    rw = torch.randn(9, 8, generator=g)
    fx["relpos_q"], fx["relpos_attn"], fx["relpos_rh"], fx["relpos_rw"] = q.numpy(), attn.numpy(), rh.numpy(), rw.numpy()
    fx["relpos_out"] = add_decomposed_rel_pos(attn, q, rh, rw, (5, 5), (5, 5)).numpy()

    # --- MLPBlock (GELU erf) and LayerNorm2d
    mlp = MLPBlock(16, 64).eval()
    load_synth(mlp, "small.mlp.")
    x = torch.randn(5, 16, generator=g) * 2
    fx["mlp_x"] = x.numpy()
    with torch.no_grad():
        fx["mlp_y"] = mlp(x).numpy()
    ln2 = LayerNorm2d(8).eval()
    load_synth(ln2, "small.ln2d.norm.")
    x = torch.randn(2, 8, 3, 3, generator=g)
    fx["ln2d_x"] = x.numpy()
    with torch.no_grad():
        fx["ln2d_y"] = ln2(x).numpy()

    # --- two-way transformer + heads at reduced width (dim 32, 4 heads, grid 6, 5 tokens)
    tw = M.TwoWayTransformer(depth=2, embedding_dim=32, mlp_dim=64, num_heads=4).eval()
    load_synth(tw, "small.dec.mask_decoder.transformer.")
    src = torch.randn(2, 32, 6, 6, generator=g)
    pos = torch.randn(1, 32, 6, 6, generator=g)
    tok = torch.randn(2, 5, 32, generator=g)
    with torch.no_grad():
        qo, ko = tw(src, pos.expand(2, -1, -1, -1), tok)
    fx["tw_src"], fx["tw_pos"], fx["tw_tok"] = src.numpy(), pos.numpy(), tok.numpy()
    fx["tw_queries"], fx["tw_keys"] = qo.numpy(), ko.numpy()
    head = MLP(32, 32, 8, 3).eval()
    load_synth(head, "small.dec.mask_decoder.class_embed.")
    with torch.no_grad():
        fx["head_y"] = head(qo).numpy()

    # --- dense PE
    pe = M.PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024),
                         mask_in_chans=16).eval()
    load_synth(pe, "prompt_encoder.")
    with torch.no_grad():
        fx["dense_pe_sample"] = sample(pe.get_dense_pe(), 8192)

    np.savez_compressed(os.path.join(out, "small_ops.npz"), **fx)
    print("wrote small_ops.npz", sum(v.nbytes for v in fx.values()) // 1024, "KiB")


# ----------------------------------------------------------------------------
def build_ref_model(model_type: str):
    """Reference encoder/decoder/prompt encoder at the factory's dims (build_sam.py:260-309)."""
    M = ref_modeling()
    d = synth.MODEL_DIMS[model_type]
    enc = M.ImageEncoderViT(
        depth=d.depth, embed_dim=d.embed_dim, img_size=1024, mlp_ratio=4,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=d.num_heads, patch_size=16,
        qkv_bias=True, use_rel_pos=True, global_attn_indexes=list(d.global_attn_indexes),
        window_size=14, out_chans=256).eval()
    dec = M.MaskDecoder(
        num_multimask_outputs=50,
        transformer=M.TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
        transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256).eval()
    pe = M.PromptEncoder(embed_dim=256, image_embedding_size=(64, 64), input_image_size=(1024, 1024),
                         mask_in_chans=16).eval()
    return enc, dec, pe


def gen_e2e(out: str, model_type: str, n_tiles: int, first_tile: int) -> None:
    t0 = time.time()
    enc, dec, pe = build_ref_model(model_type)
    W = {}
    W.update(load_synth(enc, "image_encoder."))
    W.update(load_synth(dec, "mask_decoder."))
    W.update(load_synth(pe, "prompt_encoder."))
    # the state-dict names the reference produces must equal the build's enumeration
    names = synth.weight_shapes(model_type)
    assert set(names) == set(W), (set(names) ^ set(W))
    for k, shp in names.items():
        assert tuple(W[k].shape) == tuple(shp), k
    print(f"[{model_type}] built + loaded in {time.time() - t0:.1f}s")

    x = torch.from_numpy(synth.make_batch(first_tile, n_tiles))
    hfc = O.hfc_fft(x)                               # Grayscale is torchvision -> oracle's own (unpinned)
    taps = {}
    hooks = []
    for i, blk in enumerate(enc.blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"block{i}", o.detach())))
    stem = {}
    hooks.append(enc.hfc_attn.register_forward_hook(lambda m, a, o: stem.__setitem__("hfc_attn", o.detach())))
    # token stream as it enters blocks[0]: patch embed + pos_embed + HFC adaptor output (image_encoder.py:124-131)
    hooks.append(enc.blocks[0].register_forward_pre_hook(lambda m, a: stem.__setitem__("stem", a[0].detach())))
    t1 = time.time()
    with torch.no_grad():
        emb = enc(x, hfc)
        res = dec(image_embeddings=emb, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=None,
                  dense_prompt_embeddings=None, multimask_output=False, hfc_embed=None)
    print(f"[{model_type}] reference forward {n_tiles} tile(s): {time.time() - t1:.1f}s")
    for h in hooks:
        h.remove()

    fx = {
        "model_type": np.array(model_type),
        "first_tile": np.array(first_tile), "n_tiles": np.array(n_tiles), "weight_seed": np.array(0),
        "pred_logits": res["pred_logits"].numpy(), "pred_boxes": res["pred_boxes"].numpy(),
        "hfc_sample": sample(hfc, 8192), "hfc_stats": stats(hfc), "hfc_pinned": np.array(0),
        "emb_sample": sample(emb, 16384), "emb_stats": stats(emb),
        "hfc_attn_sample": sample(stem["hfc_attn"], 8192), "hfc_attn_stats": stats(stem["hfc_attn"]),
        "stem_sample": sample(stem["stem"], 8192), "stem_stats": stats(stem["stem"]),
    }
    for k, v in taps.items():
        fx[k + "_sample"] = sample(v, 2048)
        fx[k + "_stats"] = stats(v)
    # per-channel checksum of the embedding (256 values per tile)
    fx["emb_chan_mean"] = emb.double().mean(dim=(2, 3)).numpy()

    # the same embedding through the reference decoder loaded with the "sensitive" weight profile
    sens = {}
    for k, v in dec.state_dict().items():
        sens[k] = torch.from_numpy(synth.make_weight("mask_decoder." + k, tuple(v.shape), 0, "sensitive"))
    dec.load_state_dict(sens, strict=True)
    with torch.no_grad():
        res_s = dec(image_embeddings=emb, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=None,
                    dense_prompt_embeddings=None, multimask_output=False, hfc_embed=None)
    fx["sens_pred_logits"] = res_s["pred_logits"].numpy()
    fx["sens_pred_boxes"] = res_s["pred_boxes"].numpy()

    # PostProcess + NMS lists: reference PostProcess/nms need torchvision -> oracle's own (unpinned)
    ts = torch.tensor([[1024, 1024]] * n_tiles)
    for tag, rr in (("", res), ("sens_", res_s)):
        pp = O.postprocess(rr["pred_logits"], rr["pred_boxes"], ts)
        for b, r in enumerate(pp):
            det = O.detect(r)
            fx[f"{tag}pp{b}_scores"] = r["scores"].numpy()
            fx[f"{tag}pp{b}_labels"] = r["labels"].numpy()
            fx[f"{tag}pp{b}_boxes"] = r["boxes"].numpy()
            fx[f"{tag}pp{b}_nms_index"] = det["nms_index"].numpy()
    pp = O.postprocess(res["pred_logits"], res["pred_boxes"], ts)
    fx["postprocess_pinned"] = np.array(0)
    np.savez_compressed(os.path.join(out, f"e2e_{model_type}.npz"), **fx)
    print(f"[{model_type}] wrote e2e_{model_type}.npz", sum(v.nbytes for v in fx.values()) // 1024, "KiB",
          "scores>0.5:", [int((r['scores'] > 0.5).sum()) for r in pp],
          "kept:", [len(fx[f'pp{b}_nms_index']) for b in range(n_tiles)])


def gen_e2e_outputs(out: str, model_type: str, n_tiles: int, first_tile: int, seed: int = 0, smooth: bool = False, content: int = 1024,
                    profile: str = "baseline") -> None:
    """More tiles of the same model, outputs only (logits, boxes, the oracle-derived NMS list): the end-to-end tolerance is
    then checked on several inputs, not on one.  One tile per forward (bounded memory)."""
    enc, dec, pe = build_ref_model(model_type)
    load_synth(enc, "image_encoder.", seed, profile)
    load_synth(dec, "mask_decoder.", seed, profile)
    load_synth(pe, "prompt_encoder.", seed, profile)
    fx = {"model_type": np.array(model_type), "first_tile": np.array(first_tile), "n_tiles": np.array(n_tiles), "weight_seed": np.array(seed),
          "profile": np.array(profile)}
    peak = {}
    if profile != "baseline":      # what the profile does to the activations: per-block max |x| and rms of the residual stream
        for i, blk in enumerate(enc.blocks):
            blk.register_forward_hook(lambda m, a, o, i=i: peak.__setitem__(i, (float(o.abs().max()), float(o.pow(2).mean().sqrt()))))
    lg, bx = [], []
    for t in range(n_tiles):
        x = torch.from_numpy(synth.make_batch(first_tile + t, 1, smooth=smooth))
        if content < 1024:                           # the val pipeline's input: content in the top-left corner, zeros elsewhere (utils/misc.py:50-64)
            x[:, :, content:, :] = 0
            x[:, :, :, content:] = 0
        t1 = time.time()
        with torch.no_grad():
            emb = enc(x, O.hfc_fft(x))
            res = dec(image_embeddings=emb, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=None,
                      dense_prompt_embeddings=None, multimask_output=False, hfc_embed=None)
        print(f"[{model_type}] tile {first_tile + t}: reference forward {time.time() - t1:.1f}s")
        lg.append(res["pred_logits"].numpy())
        bx.append(res["pred_boxes"].numpy())
        if peak:
            fx[f"resid_max_rms_tile{t}"] = np.array([peak[i] for i in sorted(peak)], dtype=np.float32)
        det = O.detect(O.postprocess(res["pred_logits"], res["pred_boxes"], torch.tensor([[1024, 1024]]))[0])
        fx[f"pp{t}_nms_index"] = det["nms_index"].numpy()
    fx["pred_logits"], fx["pred_boxes"] = np.concatenate(lg, 0), np.concatenate(bx, 0)
    name = f"e2e_{model_type}_tiles{first_tile}to{first_tile + n_tiles - 1}.npz" if seed == 0 else f"e2e_{model_type}_seed{seed}.npz"
    if smooth:
        name = f"e2e_{model_type}_smooth.npz"
        fx["smooth"] = np.array(1)
    if content < 1024:
        name = f"e2e_{model_type}_padded{content}.npz"
        fx["content"] = np.array(content)
    if profile != "baseline":
        name = f"e2e_{model_type}_{profile}.npz"
    np.savez_compressed(os.path.join(out, name), **fx)
    print(f"[{model_type}] wrote {name}", sum(v.nbytes for v in fx.values()) // 1024, "KiB kept:", [len(fx[f'pp{t}_nms_index']) for t in range(n_tiles)])


def ref_postprocess():
    """The reference's own PostProcess + box_cxcywh_to_xyxy, extracted by definition node (no stand-in modules)."""
    import ast
    import types
    import torch.nn.functional as F
    from torch import nn

    def node_src(path, kind, name):
        with open(path) as f:
            src = f.read()
        for n in ast.parse(src).body:
            if isinstance(n, kind) and n.name == name:
                return ast.get_source_segment(src, n)
        raise KeyError(name)

    ns_box = {"torch": torch}
    exec(compile(node_src(os.path.join(REF, "utils", "box_ops.py"), ast.FunctionDef, "box_cxcywh_to_xyxy"), "box_ops.py", "exec"), ns_box)
    ns = {"torch": torch, "F": F, "nn": nn, "box_ops": types.SimpleNamespace(box_cxcywh_to_xyxy=ns_box["box_cxcywh_to_xyxy"])}
    exec(compile(node_src(os.path.join(REF, "build_sam.py"), ast.ClassDef, "PostProcess"), "build_sam.py", "exec"), ns)
    return ns["PostProcess"], ns_box["box_cxcywh_to_xyxy"]


def gen_postprocess(out: str) -> None:
    """A19 pinned: reference PostProcess outputs on (a) the logits / boxes of every committed end-to-end fixture, (b) seeded
    random heads with non-square target sizes, (c) a tile with nothing above the confidence threshold."""
    PP, cxcywh = ref_postprocess()
    pp = PP(confidence_threshold=0.05).eval()
    fx = {"pinned": np.array(1)}
    cases = []
    gold = out
    for name in ("e2e_vit_b.npz", "e2e_vit_l.npz", "e2e_vit_h.npz", "e2e_vit_h_tiles1to4.npz", "e2e_vit_h_seed1.npz"):
        d = np.load(os.path.join(gold, name))
        lg, bx = torch.from_numpy(d["pred_logits"]), torch.from_numpy(d["pred_boxes"])
        cases.append((name[:-4], lg, bx, torch.tensor([[1024, 1024]] * lg.shape[0])))
    g = torch.Generator().manual_seed(19)
    lg = torch.randn(3, 51, 8, generator=g) * 3
    bx = torch.rand(3, 51, 4, generator=g)
    cases.append(("random", lg, bx, torch.tensor([[640, 480], [1024, 768], [333, 1000]])))
    lg0 = torch.zeros(1, 51, 8)
    lg0[..., -1] = 12.0                                     # background wins everywhere: nothing above 0.05
    cases.append(("empty", lg0, torch.rand(1, 51, 4, generator=g), torch.tensor([[1024, 1024]])))
    fx["cases"] = np.array([c[0] for c in cases])
    for tag, lg, bx, ts in cases:
        with torch.no_grad():
            res = pp({"pred_logits": lg, "pred_boxes": bx}, ts)
        fx[f"{tag}_logits"], fx[f"{tag}_boxes"], fx[f"{tag}_sizes"] = lg.numpy(), bx.numpy(), ts.numpy()
        for b, r in enumerate(res):
            fx[f"{tag}_pp{b}_scores"] = r["scores"].to(torch.float32).numpy()
            fx[f"{tag}_pp{b}_labels"] = r["labels"].to(torch.int64).numpy()
            fx[f"{tag}_pp{b}_boxes"] = r["boxes"].to(torch.float32).numpy().reshape(-1, 4)
        # the oracle restatement must agree bit for bit before the fixture is written
        mine = O.postprocess(lg, bx, ts)
        for b, r in enumerate(res):
            assert torch.equal(mine[b]["scores"], r["scores"].to(torch.float32)), (tag, b)
            assert torch.equal(mine[b]["labels"], r["labels"].to(torch.int64)), (tag, b)
            assert torch.equal(mine[b]["boxes"].reshape(-1, 4), r["boxes"].to(torch.float32).reshape(-1, 4)), (tag, b)
    x = torch.rand(5, 9, 4, generator=g)
    fx["cxcywh_in"], fx["cxcywh_out"] = x.numpy(), cxcywh(x).numpy()
    np.savez_compressed(os.path.join(out, "postprocess_ref.npz"), **fx)
    print("wrote postprocess_ref.npz", sum(v.nbytes for v in fx.values()) // 1024, "KiB; cases:", [c[0] for c in cases],
          "(oracle restatement bit-identical on all)")


def gen_coco_subset(out: str, n_images: int = 8) -> None:
    """Data fixture for the COCO bbox evaluator: the first images of the reference's own annotation file
    (coco_annotations/val.json: 6 categories, xywh boxes, area, iscrowd) with their annotations, verbatim."""
    import json
    with open("/root/reference/coco_annotations/val.json") as f:
        d = json.load(f)
    imgs = d["images"][:n_images]
    ids = {i["id"] for i in imgs}
    sub = {"info": d.get("info", {}), "images": imgs, "categories": d["categories"],
           "annotations": [a for a in d["annotations"] if a["image_id"] in ids]}
    with open(os.path.join(out, "coco_val_subset.json"), "w") as f:
        json.dump(sub, f)
    print("wrote coco_val_subset.json:", len(imgs), "images,", len(sub["annotations"]), "annotations")


def gen_resize(out: str) -> None:
    """Golden vectors of the val transform's resize (dataloader_coco.py:288 -> augmentation.py:77-133 ->
    torchvision F.resize on a PIL image = PIL.Image.resize(BILINEAR)): inputs and PIL's own outputs, small cases
    (down-scaling with the antialiasing filter at several ratios, up-scaling, one axis unchanged)."""
    from PIL import Image
    import PIL
    from oracle import pil_resize as R
    rng = np.random.default_rng(11)
    fx = {"pillow_version": np.array(PIL.__version__)}
    cases = [(150, 225, 64, 64), (40, 60, 64, 96), (182, 273, 48, 48), (64, 48, 96, 128), (114, 171, 24, 24), (100, 150, 65, 65), (64, 64, 64, 64)]
    fx["cases"] = np.array(cases, dtype=np.int32)
    for i, (h, w, size, mx) in enumerate(cases):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i == 2:                                   # smooth content too, not only noise
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(yy * 255 // h), (xx * 255 // w), ((yy + xx) % 256)], axis=-1).astype(np.uint8)
        oh, ow = R.get_size_with_aspect_ratio((w, h), size, mx)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        fx[f"in{i}"], fx[f"out{i}"] = img, ref
    np.savez_compressed(os.path.join(out, "resize_pil.npz"), **fx)
    print("wrote resize_pil.npz", sum(v.nbytes for v in fx.values()) // 1024, "KiB (Pillow", PIL.__version__ + ")")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    if a.only in ("all", "small"):
        gen_small(a.out)
    if a.only in ("all", "resize"):
        gen_resize(a.out)
    if a.only in ("postprocess",):                # after the end-to-end fixtures exist (reads their logits / boxes)
        gen_postprocess(a.out)
    if a.only in ("all", "coco"):
        gen_coco_subset(a.out)
    if a.only in ("all", "vit_b"):
        gen_e2e(a.out, "vit_b", n_tiles=2, first_tile=0)
    if a.only in ("all", "vit_l"):
        gen_e2e(a.out, "vit_l", n_tiles=1, first_tile=5)
    if a.only in ("all", "vit_h"):
        gen_e2e(a.out, "vit_h", n_tiles=1, first_tile=0)
    if a.only in ("all", "vit_h_tiles"):
        gen_e2e_outputs(a.out, "vit_h", n_tiles=4, first_tile=1)
    if a.only in ("all", "vit_h_seed1"):
        gen_e2e_outputs(a.out, "vit_h", n_tiles=2, first_tile=0, seed=1)
    if a.only in ("all", "vit_h_smooth"):
        gen_e2e_outputs(a.out, "vit_h", n_tiles=2, first_tile=0, smooth=True)
    if a.only in ("all", "vit_h_outlier"):
        gen_e2e_outputs(a.out, "vit_h", n_tiles=2, first_tile=0, profile="outlier")
    if a.only in ("all", "vit_h_padded"):
        gen_e2e_outputs(a.out, "vit_h", n_tiles=2, first_tile=0, smooth=True, content=768)


if __name__ == "__main__":
    main()
