"""End-to-end parity (-m gpu): the HIP path through the drop-in Python API against
  (a) the golden fixtures produced by the reference's own modules (tests/golden, fp32 CPU), and
  (b) the CPU oracle run live on the same seeded inputs (small / cheap cases only).

Tolerances.  north_star asks for logits within 1e-3 relative of the reference CPU forward and
identical box indices after NMS.  "Relative" is measured as ||x - ref||_2 / ||ref||_2 over the
(B,51,8) logits.  Operand precision sets what is reachable (DESIGN.md "Precision"):
  fp16 operands (the default): 1e-3 on logits on both weight profiles and both weight seeds, identical NMS indices;
  bf16 operands (transformer blocks in bf16; stem, HFC adaptor and neck always run fp16 operands):
                 1e-3 on logits on the baseline profile with weight seed 0 (measured 7.2-8.3e-4 ViT-H on five tiles, 4.7e-4
                 ViT-B) but 2.0e-3 with seed 1; identical NMS indices on both; the embedding itself is checked at 5e-3
                 (8-bit mantissas give ~3e-3 per GEMM).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wm_oracle as O          # checker only
from wildlifemapper_amd import synth
from wildlifemapper_amd.engine import split_records
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM
from wildlifemapper_amd.segment_anything.utils.misc import NestedTensor, nested_tensor_from_tensor_list
import gpu_util as G

LOGIT_TOL = {"fp16": 1e-3, "bf16": 1e-3}
# what the tests ASSERT, per model below north_star's 1e-3 so that the margin itself is checked and a regression of it
# shows.  Measured: bf16 ViT-B 3.0-4.7e-4, ViT-H 8.2e-4, ViT-L 9.2e-4 (every block's 8-bit-mantissa operand rounding adds
# alike; WM_FP16_TAIL / fp16 mode trade throughput for margin: DESIGN.md section 3); fp16 2.4e-4 / 7.8e-5.
LOGIT_ASSERT = {"fp16": {"vit_b": 2e-4, "vit_l": 4e-4, "vit_h": 4e-4}, "bf16": {"vit_b": 6e-4, "vit_l": 9.8e-4, "vit_h": 9e-4}}
EMB_TOL = {"fp16": 2e-3, "bf16": 5e-3}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")


def _sample(t, n):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].float().cpu().numpy()


_MODELS = {}


def _model(mt, prec):
    """One model per type, kept for the whole module (ViT-H weights take ~30 s to synthesise);
    switching precision re-packs the weights into a fresh native handle."""
    if mt not in _MODELS:
        for k in list(_MODELS):                               # one model type resident at a time
            _MODELS.pop(k)[0]._hub.close()
        sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(mt).items()}
        sam, crit, post = sam_model_registry[mt](None, None)
        m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
        m.load_state_dict(sd, strict=True)
        _MODELS[mt] = (m, post["bbox"])
    m, post = _MODELS[mt]
    m._hub.set_precision(prec)
    return m, post


# ---------------------------------------------------------------------------
# FFT high-pass (A2)
# ---------------------------------------------------------------------------
def test_hfc_fft_matches_oracle():
    m, _ = _model("vit_b", "fp16")
    x = torch.from_numpy(synth.make_batch(3, 2, smooth=True))
    got = m.fft(NestedTensor(x.to(G.dev()), None)).cpu()
    ref = O.hfc_fft(x)
    assert got.shape == (2, 1, 1024, 1024)
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # size-independent property: a constant image has no high-frequency component
    flat = torch.full((1, 3, 1024, 1024), 0.37, device=G.dev())
    assert m.fft(flat).abs().max().item() < 1e-5
    # linearity of the pre-|.| map shows up as homogeneity: fft(2x) = 2 fft(x)
    xd = x[:1].to(G.dev())
    assert torch.allclose(m.fft(2 * xd), 2 * m.fft(xd), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------
# PostProcess + NMS (A19, A20)
# ---------------------------------------------------------------------------
def _check_postprocess(post, logits, boxes, ts):
    dev = G.dev()
    out = {"pred_logits": logits.to(dev), "pred_boxes": boxes.to(dev)}
    got = post(out, ts.to(dev))
    got_nms = post.forward_with_nms(out, ts.to(dev))
    ref = O.postprocess(logits, boxes, ts)
    for b, r in enumerate(ref):
        g = got[b]
        assert g["scores"].shape == r["scores"].shape
        np.testing.assert_allclose(g["scores"].cpu().numpy(), r["scores"].numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_array_equal(g["labels"].cpu().numpy(), r["labels"].numpy())
        np.testing.assert_allclose(g["boxes"].cpu().numpy(), r["boxes"].numpy(), rtol=1e-6, atol=1e-4)
        det = O.detect(r)
        np.testing.assert_array_equal(got_nms[b]["nms_index"].cpu().numpy(), det["nms_index"].numpy())
        np.testing.assert_allclose(got_nms[b]["boxes"].cpu().numpy(), det["boxes"].numpy(), rtol=1e-6, atol=1e-4)


def test_postprocess_nms_random_and_edges():
    _, post = _model("vit_b", "fp16")
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(4, 51, 8, generator=g) * 3
    boxes = torch.rand(4, 51, 4, generator=g) * torch.tensor([1, 1, 0.4, 0.4])
    logits[1, :, 7] += 20            # tile 1: background wins everywhere -> nothing above 0.05 (empty result)
    logits[2, :, 3] += 12            # tile 2: every slot a confident detection -> NMS does real work
    boxes[3, 10] = boxes[3, 11]      # identical boxes, near-tied scores
    logits[3, 11] = logits[3, 10]
    ts = torch.tensor([[1024, 1024], [768, 512], [1024, 1024], [4000, 6000]])
    _check_postprocess(post, logits, boxes, ts)
    assert len(post({"pred_logits": logits.to(G.dev()), "pred_boxes": boxes.to(G.dev())}, ts.to(G.dev()))[1]["scores"]) == 0


def test_postprocess_vs_reference_fixture(golden_dir):
    """A19 against the reference's OWN PostProcess outputs (tests/golden/postprocess_ref.npz, pinned=1: the class body of
    build_sam.py:212-258 run by oracle/gen_golden.py --only postprocess): postprocess_nms_kernel's scores / labels / scaled
    boxes on every case of the fixture, including non-square target sizes and the empty result."""
    _, post = _model("vit_b", "fp16")
    fx = np.load(os.path.join(golden_dir, "postprocess_ref.npz"))
    assert int(fx["pinned"]) == 1
    dev = G.dev()
    for tag in fx["cases"]:
        tag = str(tag)
        lg, bx, ts = (torch.from_numpy(fx[f"{tag}_{k}"]) for k in ("logits", "boxes", "sizes"))
        got = post({"pred_logits": lg.to(dev), "pred_boxes": bx.to(dev)}, ts.to(dev))
        for b, g in enumerate(got):
            want_s, want_l, want_b = fx[f"{tag}_pp{b}_scores"], fx[f"{tag}_pp{b}_labels"], fx[f"{tag}_pp{b}_boxes"]
            assert g["scores"].shape[0] == want_s.shape[0], (tag, b)
            np.testing.assert_allclose(g["scores"].cpu().numpy(), want_s, rtol=2e-6, atol=1e-7)
            np.testing.assert_array_equal(g["labels"].cpu().numpy(), want_l)
            np.testing.assert_allclose(g["boxes"].cpu().numpy().reshape(-1, 4), want_b, rtol=1e-6, atol=1e-4)


# ---------------------------------------------------------------------------
# full model vs golden fixtures
# ---------------------------------------------------------------------------
def _run_vs_golden(mt, prec, golden_dir):
    fx = np.load(os.path.join(golden_dir, f"e2e_{mt}.npz"))
    n = int(fx["n_tiles"])
    m, post = _model(mt, prec)
    x = torch.from_numpy(synth.make_batch(int(fx["first_tile"]), n)).to(G.dev())
    hub = m._hub
    hub.handle(x.device, n)
    report = {}
    # encoder taps: stem, a middle block, the last block
    depth = synth.MODEL_DIMS[mt].depth
    hfc = m.fft(x)
    np.testing.assert_allclose(_sample(hfc, 8192), fx["hfc_sample"], atol=3e-5)
    # the stem directly (image_encoder.py:124-131): patch embed + pos_embed (tap -3), + the HFC adaptor's output (tap -1 =
    # the input of blocks[0]); their difference is CrossAttentionHfcPatch's output.  The stem always runs fp16 operands.
    hub.set_tap(-3)
    m.image_encoder(x, hfc)
    tokbase = hub.read_tap(n)
    hub.set_tap(-1)
    m.image_encoder(x, hfc)
    stem = hub.read_tap(n)
    ref = fx["stem_sample"]
    report["stem"] = np.linalg.norm(_sample(stem, 8192) - ref) / np.linalg.norm(ref)
    assert report["stem"] < 1e-3, report["stem"]
    ref = fx["hfc_attn_sample"]
    report["hfc_attn"] = np.linalg.norm(_sample(stem - tokbase, 8192) - ref) / np.linalg.norm(ref)
    assert report["hfc_attn"] < 2e-3, report["hfc_attn"]
    rms = float(stem.double().pow(2).mean().sqrt().item())
    assert abs(rms - fx["stem_stats"][2]) < 1e-3 * fx["stem_stats"][2], (rms, fx["stem_stats"][2])
    taps = sorted(set(range(0, depth, max(1, depth // 8))) | {depth // 2, depth - 1})
    for which in taps:
        hub.set_tap(which)
        emb = m.image_encoder(x, hfc)
        tap = hub.read_tap(n)
        ref = fx[f"block{which}_sample"]
        err = np.linalg.norm(_sample(tap, 2048) - ref) / np.linalg.norm(ref)
        report[f"block{which}"] = err
        assert err < EMB_TOL[prec], (which, err)
        rms = float(tap.double().pow(2).mean().sqrt().item())              # full-tensor statistic, not only the sample
        assert abs(rms - fx[f"block{which}_stats"][2]) < EMB_TOL[prec] * fx[f"block{which}_stats"][2], (which, rms)
    hub.set_tap(-2)
    ref = fx["emb_sample"]
    err = np.linalg.norm(_sample(emb, 16384) - ref) / np.linalg.norm(ref)
    report["embedding"] = err
    assert err < EMB_TOL[prec], err
    np.testing.assert_allclose(emb.double().mean(dim=(2, 3)).cpu().numpy(), fx["emb_chan_mean"], atol=EMB_TOL[prec])

    # whole path in one native call
    out = m.detect(NestedTensor(x, None), torch.tensor([[1024, 1024]] * n))
    lg, bx = out["pred_logits"].cpu().numpy(), out["pred_boxes"].cpu().numpy()
    lerr = np.linalg.norm(lg - fx["pred_logits"]) / np.linalg.norm(fx["pred_logits"])
    berr = np.abs(bx - fx["pred_boxes"]).max()
    report["logits"], report["boxes_maxabs"] = lerr, berr
    print(f"[{mt}/{prec}] " + " ".join(f"{k}={v:.2e}" for k, v in report.items()))
    print(f"[{mt}/{prec}] logits margin: {lerr:.2e} of the 1e-3 bar (asserted < {LOGIT_ASSERT[prec][mt]:.1e})")
    assert lerr < LOGIT_ASSERT[prec][mt], lerr
    assert berr < 5 * LOGIT_TOL[prec], berr

    # NMS indices: identical to the reference-derived list (fp16); reported with margins otherwise
    rec = split_records(out["records"].cpu())
    same = []
    for b in range(n):
        flags, rank = rec["flags"][b], rec["nms_rank"][b]
        cand = (flags & 2) != 0
        pos = torch.cumsum(cand.long(), 0) - 1
        slots = torch.nonzero((flags & 4) != 0).flatten()
        slots = slots[torch.argsort(rank[slots])]
        same.append(np.array_equal(pos[slots].numpy(), fx[f"pp{b}_nms_index"]))
    print(f"[{mt}/{prec}] NMS index lists identical: {same}")
    assert all(same)
    return out


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_vit_b_vs_reference_golden(prec, golden_dir):
    _run_vs_golden("vit_b", prec, golden_dir)


def _sensitive(mt, prec, golden_dir, n):
    """Stricter profile: peaky token->image attention amplifies encoder error ~4x into the logits."""
    fx = np.load(os.path.join(golden_dir, f"e2e_{mt}.npz"))
    m, _ = _model(mt, prec)
    sens = {k: torch.from_numpy(synth.make_weight(k, s, 0, "sensitive")) for k, s in synth.weight_shapes(mt).items()
            if k.startswith("mask_decoder.")}
    base = {k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith("mask_decoder.")}
    try:
        m.load_state_dict(sens, strict=False)
        x = torch.from_numpy(synth.make_batch(0, n)).to(G.dev())
        out = m.detect(NestedTensor(x, None), torch.tensor([[1024, 1024]] * n))
        lg = out["pred_logits"].cpu().numpy()
        err = np.linalg.norm(lg - fx["sens_pred_logits"]) / np.linalg.norm(fx["sens_pred_logits"])
        rec = split_records(out["records"].cpu())
        same = [_nms_positions(rec, b) == fx[f"sens_pp{b}_nms_index"].tolist() for b in range(n)]
        print(f"[{mt}/{prec}/sensitive] logits={err:.2e} NMS identical: {same}")
        return err, same
    finally:
        m.load_state_dict(base, strict=False)


def _nms_positions(rec, b):
    """Kept boxes as positions inside the score-filtered candidate list, in NMS order (visualize_prediction.py:150-154)."""
    flags, rank = rec["flags"][b], rec["nms_rank"][b]
    pos = torch.cumsum(((flags & 2) != 0).long(), 0) - 1
    slots = torch.nonzero((flags & 4) != 0).flatten()
    slots = slots[torch.argsort(rank[slots])]
    return pos[slots].tolist()


def test_vit_b_sensitive_profile_fp16(golden_dir):
    err, same = _sensitive("vit_b", "fp16", golden_dir, 2)
    assert err < 1e-3, err                 # measured 4.4e-4
    assert all(same)


# ---------------------------------------------------------------------------
# drop-in surface
# ---------------------------------------------------------------------------
def test_dropin_modules_and_batch_invariance():
    m, post = _model("vit_b", "fp16")
    x = torch.from_numpy(synth.make_batch(10, 3)).to(G.dev())
    hfc = m.fft(x)
    emb = m.image_encoder(x, hfc)                                   # ImageEncoderViT.forward(x, x_hfc)
    assert emb.shape == (3, 256, 64, 64) and emb.dtype == torch.float32
    out = m.mask_decoder(image_embeddings=emb, image_pe=m.prompt_encoder.get_dense_pe(), sparse_prompt_embeddings=None,
                         dense_prompt_embeddings=None, multimask_output=False, hfc_embed=None)
    assert out["pred_logits"].shape == (3, 51, 8) and out["pred_boxes"].shape == (3, 51, 4)
    fused = m(NestedTensor(x, None), None)
    assert torch.equal(fused["pred_logits"], out["pred_logits"]) and torch.equal(fused["pred_boxes"], out["pred_boxes"])
    # tiles are independent: a tile's result does not depend on its batch neighbours
    single = m(NestedTensor(x[1:2].contiguous(), None), None)
    assert torch.equal(single["pred_logits"][0], fused["pred_logits"][1])
    # NestedTensor collation (utils/misc.py:46-67): short image, top-left aligned, zero padded
    nt = nested_tensor_from_tensor_list([x[0, :, :768, :700].cpu(), x[1].cpu()])
    assert nt.tensors.shape == (2, 3, 1024, 1024) and bool(nt.mask[0, 800, 0]) and not bool(nt.mask[0, 0, 0])
    res = post(m(nt.to(G.dev()), None), torch.tensor([[1024, 1024]] * 2, device=G.dev()))
    assert set(res[0]) == {"scores", "labels", "boxes"}


def test_partial_weight_reupload_matches_fresh_handle():
    """A parameter changed in place after the first forward (here a window block's and a global block's qkv bias, plus a LayerNorm
    gamma) is re-uploaded into the SAME native handle; everything the handle derives from weights -- packed copies, folded LayerNorm
    weights, the 16-bit qkv-bias rows the window attention reads for padded tokens (cached by device address) -- must follow.  The
    result has to equal a fresh handle's bit for bit."""
    m, _ = _model("vit_b", "fp16")
    x = torch.from_numpy(synth.make_batch(4, 2)).to(G.dev())
    hub = m._hub
    first = m(NestedTensor(x, None), None)["pred_logits"].clone()
    blocks = m.image_encoder.blocks
    touched = [blocks[0].attn.qkv.bias, blocks[2].attn.qkv.bias, blocks[1].norm1.weight]
    saved = [t.detach().clone() for t in touched]
    try:
        with torch.no_grad():
            touched[0].add_(0.25); touched[1].mul_(-1.5); touched[2].mul_(1.1)
        again = m(NestedTensor(x, None), None)["pred_logits"].clone()          # partial re-upload into the live handle
        assert not torch.equal(again, first)
        hub.close()                                                             # fresh handle, full upload of the same weights
        fresh = m(NestedTensor(x, None), None)["pred_logits"].clone()
        assert torch.equal(again, fresh)
    finally:
        with torch.no_grad():
            for t, s0 in zip(touched, saved): t.copy_(s0)
        hub.close()
    restored = m(NestedTensor(x, None), None)["pred_logits"]
    assert torch.equal(restored, first)


def test_folded_layernorm_off_switch_and_tile_independence():
    """WM_LN_FOLD=0 (hub.fold_ln = False) runs the blocks' LayerNorm as its own kernel: both paths meet the reference fixture,
    within each a tile's bits do not depend on its batch neighbours, and the two differ only within the operand rounding."""
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_vit_b.npz"))
    m, _ = _model("vit_b", "fp16")
    x = torch.from_numpy(synth.make_batch(0, 4)).to(G.dev())
    hub = m._hub
    fold_was = hub.fold_ln
    outs = {}
    try:
        for mode in (True, False):
            hub.fold_ln = mode
            hub.close()
            full = m(NestedTensor(x, None), None)
            one = m(NestedTensor(x[1:2].contiguous(), None), None)
            assert torch.equal(one["pred_logits"][0], full["pred_logits"][1]), mode
            lg = full["pred_logits"][:2].cpu().numpy()
            err = float(np.linalg.norm(lg - fx["pred_logits"]) / np.linalg.norm(fx["pred_logits"]))
            assert err < LOGIT_ASSERT["fp16"]["vit_b"], (mode, err)
            outs[mode] = full["pred_logits"].cpu()
    finally:
        hub.fold_ln = fold_was
        hub.close()
    assert not torch.equal(outs[True], outs[False])
    assert G.rel_l2(outs[True], outs[False]) < 3e-4


def test_evaluate_harness(golden_dir):
    """inference.evaluate (inference.py:30-89): loader of (NestedTensor, targets) -> (stats, coco_evaluator) with
    stats['coco_eval_bbox'] (12 COCO numbers).  Ground truth = the detections PostProcess derives from the reference's own
    logits (golden fixture), so the HIP path must score AP = 1 against it; one image gets no ground truth at all."""
    from types import SimpleNamespace
    from wildlifemapper_amd.inference import evaluate
    from wildlifemapper_amd.segment_anything.build_sam import InferenceCriterion
    fx = np.load(os.path.join(golden_dir, "e2e_vit_b.npz"))
    m, post = _model("vit_b", "fp16")
    x = torch.from_numpy(synth.make_batch(0, 2))
    ref = O.postprocess(torch.from_numpy(fx["pred_logits"]), torch.from_numpy(fx["pred_boxes"]), torch.tensor([[1024, 1024]] * 2))
    anns, loader = [], []
    for i in range(2):
        tgt = [{"image_id": torch.tensor([100 + i]), "orig_size": torch.tensor([1024, 1024])}]
        loader.append((nested_tensor_from_tensor_list([x[i]]), tgt))
        for b, l in zip(ref[i]["boxes"].numpy(), ref[i]["labels"].numpy()):
            anns.append({"id": len(anns) + 1, "image_id": 100 + i, "category_id": int(l), "iscrowd": 0,
                         "bbox": [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])], "area": float((b[2] - b[0]) * (b[3] - b[1]))})
    base_ds = {"images": [{"id": 100}, {"id": 101}], "categories": [{"id": c} for c in range(7)], "annotations": anns}
    stats, ev = evaluate(m, InferenceCriterion(), {"bbox": post}, loader, base_ds, G.dev(), SimpleNamespace(batch_size=1))
    assert stats["images"] == 2 and ev.img_ids == [100, 101]
    assert len(stats["coco_eval_bbox"]) == 12 and stats["coco_eval_bbox"] == ev.coco_eval["bbox"].stats.tolist()
    print("[evaluate] coco_eval_bbox =", [round(v, 4) for v in stats["coco_eval_bbox"]])
    assert stats["coco_eval_bbox"][0] > 0.99 and stats["coco_eval_bbox"][1] > 0.999        # fp16 path vs the fp32 reference's own detections
    assert stats["detections"] == sum(len(r["scores"]) for r in ref)
    # no ground truth handed over: the reference's `coco_evaluator is None` branches
    stats2, ev2 = evaluate(m, None, {"bbox": post}, loader, None, G.dev(), SimpleNamespace(batch_size=1))
    assert ev2 is None and "coco_eval_bbox" not in stats2 and stats2["images"] == 2


def test_input_pipeline_bit_exact():
    """uint8 HWC -> normalised, zero-padded fp32 tile: bit-exact with the numpy pipeline (synth.normalize_tile)."""
    from wildlifemapper_amd.preprocess import tiles_from_u8
    full = np.stack([synth.make_tile_u8(t) for t in (0, 1)])
    got = tiles_from_u8(torch.from_numpy(full).to(G.dev())).cpu().numpy()
    want = np.stack([synth.normalize_tile(u) for u in full])
    assert np.array_equal(got, want)
    small = full[:, :768, :700]                                   # short image: top-left aligned, zero padded
    got = tiles_from_u8(torch.from_numpy(np.ascontiguousarray(small)).to(G.dev())).cpu().numpy()
    assert np.array_equal(got[:, :, :768, :700], want[:, :, :768, :700])
    assert not got[:, :, 768:, :].any() and not got[:, :, :, 700:].any()


def test_cpu_tensor_fails_loudly():
    m, _ = _model("vit_b", "fp16")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(NestedTensor(torch.zeros(1, 3, 1024, 1024), None), None)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_vit_l_vs_reference_golden(prec, golden_dir):
    """The registry's third entry (build_sam.py:30-41: 1024 wide, 24 blocks, global blocks 5/11/17/23; 256-column GEMM tiles,
    4-tile LayerNorm, hd 64 attention) against tests/golden/e2e_vit_l.npz: the reference's own modules on tile 5
    (oracle/gen_golden.py --only vit_l): stem, HFC adaptor, 9 block taps, embedding, logits, boxes, NMS index list."""
    _run_vs_golden("vit_l", prec, golden_dir)


def test_vit_l_vs_oracle():
    """Same model against the CPU oracle run live on the same tile (bf16 mode, the same 1e-3 bar): ties the oracle, the
    fixture and the HIP path together for the third registry entry."""
    m, post = _model("vit_l", "bf16")
    x = torch.from_numpy(synth.make_batch(5, 1))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = O.model_forward(x, sd, O.OracleCfg.from_model_type("vit_l"))
    with torch.no_grad():
        out = m.detect(x.to(G.dev()))
    lerr = G.rel_l2(out["pred_logits"].cpu(), ref["pred_logits"])
    berr = (out["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max().item()
    print(f"[vit_l/bf16] logits={lerr:.2e} boxes_maxabs={berr:.2e}")
    assert lerr < LOGIT_ASSERT["bf16"]["vit_l"], lerr
    assert berr < 5 * LOGIT_TOL["bf16"], berr
    det = O.detect(O.postprocess(ref["pred_logits"], ref["pred_boxes"], torch.tensor([[1024, 1024]]))[0])
    rec = split_records(out["records"].cpu())
    flags, rank = rec["flags"][0], rec["nms_rank"][0]
    pos = torch.cumsum(((flags & 2) != 0).long(), 0) - 1
    slots = torch.nonzero((flags & 4) != 0).flatten()
    slots = slots[torch.argsort(rank[slots])]
    assert pos[slots].tolist() == det["nms_index"].tolist()


# ViT-H last: it replaces the resident ViT-B model
@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_vit_h_vs_reference_golden(prec, golden_dir):
    _run_vs_golden("vit_h", prec, golden_dir)


def test_vit_h_sensitive_profile_fp16(golden_dir):
    err, same = _sensitive("vit_h", "fp16", golden_dir, 1)
    assert err < 1e-3, err                 # CPU operand-rounding emulation: 7e-4
    assert all(same)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_vit_h_batches_golden_tile_and_bit_identity(prec, golden_dir):
    """The configurations bench.py times (BASELINE.json configs[1] B = 4, configs[2] B = 16): tile 0 of the batch is the
    golden tile, checked against the reference-generated fixture; a tile's result is bit-identical whatever its batch
    (INTEGRATION.md), although B = 1 takes the half-width GEMM kernel for proj / lin2 and B >= 4 the staggered 256 x 320
    kernel with the LDS-DMA residual epilogue -- which instance ran is asserted from the library's launch counters."""
    from wildlifemapper_amd import _native as Nn
    fx = np.load(os.path.join(golden_dir, "e2e_vit_h.npz"))
    assert int(fx["n_tiles"]) == 1
    first = int(fx["first_tile"])
    m, _ = _model("vit_h", prec)
    x16 = torch.from_numpy(synth.make_batch(first, 16)).to(G.dev())
    ts = torch.tensor([[1024, 1024]] * 16)
    outs, variants = {}, {}
    for B in (16, 5, 4, 1):                                               # 5: an odd batch (80 row tiles: rounds that do not fill)
        xb = x16[:B].contiguous()
        m.detect(NestedTensor(xb, None), ts[:B])                      # weights packed / handle sized before counting
        Nn.gemm_variant_counts(reset=True)
        outs[B] = {k: v.cpu() for k, v in m.detect(NestedTensor(xb, None), ts[:B]).items()}
        torch.cuda.synchronize()
        variants[B] = {k: v for k, v in Nn.gemm_variant_counts().items() if v}
        print(f"[vit_h/{prec}] B={B} GEMM instances: {variants[B]}")
    depth = synth.MODEL_DIMS["vit_h"].depth
    for B in (16, 4):
        v = variants[B]
        # proj + lin2 of every block: the split-stream producer (residual planes in and out, row statistics; folded LayerNorm,
        # the default in both 16-bit modes); the stem's proj_back starts the planes where its fp16 type is also block 0's
        if prec == "fp16":
            assert v.get("v5_320_split", 0) == 2 * depth and v.get("v5_320_foldp", 0) == 1, v
        else:                                                            # bf16 blocks keep the LayerNorm kernel and the fp32 stream by default
            assert v.get("v5_320_res", 0) >= 2 * depth and v.get("v5_320_split", 0) == 0, v
        assert v.get("v5_320", 0) >= 2 * depth, v                      # qkv + lin1 of every block
        assert "v2_160" not in v and "v1_128" not in v, v
    assert variants[1].get("v2_160", 0) >= 2 * depth and not any(variants[1].get(k, 0) for k in ("v5_320_res", "v5_320_foldp", "v5_320_split")), variants[1]
    for B in (16, 5, 4, 1):
        lg = outs[B]["pred_logits"][:1].numpy()
        lerr = np.linalg.norm(lg - fx["pred_logits"]) / np.linalg.norm(fx["pred_logits"])
        berr = np.abs(outs[B]["pred_boxes"][:1].numpy() - fx["pred_boxes"]).max()
        print(f"[vit_h/{prec}] B={B} golden tile: logits={lerr:.2e} boxes_maxabs={berr:.2e}")
        assert lerr < LOGIT_ASSERT[prec]["vit_h"], (B, lerr)
        assert berr < 5 * LOGIT_TOL[prec], (B, berr)
        rec = split_records(outs[B]["records"])
        assert _nms_positions(rec, 0) == fx["pp0_nms_index"].tolist(), B
    for B in (5, 4, 1):
        for k in ("pred_logits", "pred_boxes", "records"):
            # records hold int32 fields behind a float32 view (nms_rank = -1 reads as NaN): compare the bits
            assert torch.equal(outs[B][k].view(torch.int32), outs[16][k][:B].view(torch.int32)), (B, k)
    # every tile of the batch produced detections of its own (no tile silently copied or skipped)
    lg16 = outs[16]["pred_logits"]
    assert all(not torch.equal(lg16[i], lg16[j]) for i in range(16) for j in range(i))


# ---------------------------------------------------------------------------
# fp8 (BASELINE.json configs[4]: "ViT-H fp8 MFMA weights/activations, batch=16 ... tolerance re-stated")
# ---------------------------------------------------------------------------
# Re-stated tolerance: e4m3 carries 3 mantissa bits (relative rounding error up to 2^-4 per operand
# element), so the fp32 reference is matched to about 1e-2 on the logits instead of 1e-3, measured first with the CPU
# emulation (oracle cfg.block_fp8: 9.0e-3 ViT-B, 2.95e-2 ViT-H; the GPU path reproduces both to 2 digits).  Asserted:
# logits within 5e-2 relative of the reference fixture, boxes within 3e-2 absolute, detections (score cut + NMS) scoring
# mAP50 >= 0.8 against the reference's own detections (box jitter of a few pixels costs the high-IoU thresholds of
# mAP@[.5:.95], which is printed); and the GPU path within 5e-3 of the CPU emulation of the same arithmetic (ViT-B).
FP8_LOGIT_TOL, FP8_BOX_TOL, FP8_MAP50_TOL = 5e-2, 3e-2, 0.8


def _dets_from_records(rec, b):
    kept = (rec["flags"][b] & 4) != 0
    order = torch.argsort(rec["nms_rank"][b][kept])
    return {"boxes": rec["boxes"][b][kept][order].numpy(), "scores": rec["scores"][b][kept][order].numpy(),
            "labels": rec["labels"][b][kept][order].numpy()}


def _fp8_vs_golden(mt, golden_dir, batch):
    from wildlifemapper_amd import _native as Nn
    from wildlifemapper_amd.coco_eval import map_vs_reference
    fx = np.load(os.path.join(golden_dir, f"e2e_{mt}.npz"))
    n = int(fx["n_tiles"])
    m, _ = _model(mt, "fp8")
    x = torch.from_numpy(synth.make_batch(int(fx["first_tile"]), max(batch, n))).to(G.dev())
    ts = torch.tensor([[1024, 1024]] * x.shape[0])
    m.detect(NestedTensor(x, None), ts)
    Nn.gemm_variant_counts(reset=True)
    out = {k: v.cpu() for k, v in m.detect(NestedTensor(x, None), ts).items()}
    torch.cuda.synchronize()
    var = {k: v for k, v in Nn.gemm_variant_counts().items() if v}
    depth = synth.MODEL_DIMS[mt].depth
    n_bf16 = int(os.environ.get("WM_FP8_BF16_HEAD", 0)) + int(os.environ.get("WM_FP8_BF16_TAIL", 0))
    # qkv, proj, lin1, lin2 of every fp8 block on the fp8 MFMA; proj / lin2 as the instance that keeps the stream as row-major planes
    rows = int(os.environ.get("WM_FP8_ROWS", 1)) != 0
    assert var.get("fp8_256", 0) + var.get("fp8_256_planes", 0) == 4 * (depth - n_bf16), var
    assert var.get("fp8_256_planes", 0) == (2 * (depth - n_bf16) if rows else 0), var
    lg = out["pred_logits"][:n].numpy()
    lerr = np.linalg.norm(lg - fx["pred_logits"]) / np.linalg.norm(fx["pred_logits"])
    berr = np.abs(out["pred_boxes"][:n].numpy() - fx["pred_boxes"]).max()
    rec = split_records(out["records"])
    pred = {b: _dets_from_records(rec, b) for b in range(n)}
    ref_pp = O.postprocess(torch.from_numpy(fx["pred_logits"]), torch.from_numpy(fx["pred_boxes"]), torch.tensor([[1024, 1024]] * n))
    gt = {}
    for b in range(n):
        d = O.detect(ref_pp[b])
        gt[b] = {"boxes": d["boxes"].numpy(), "scores": d["scores"].numpy(), "labels": d["labels"].numpy()}
    mp = map_vs_reference(pred, gt)
    print(f"[{mt}/fp8] B={x.shape[0]} logits={lerr:.2e} boxes_maxabs={berr:.2e} mAP / mAP50 vs reference detections={mp['mAP']:.3f} / {mp['mAP50']:.3f} "
          f"(kept {[len(pred[b]['scores']) for b in range(n)]} vs {[len(gt[b]['scores']) for b in range(n)]})  GEMM instances {var}")
    assert lerr < FP8_LOGIT_TOL, lerr
    assert berr < FP8_BOX_TOL, berr
    assert mp["mAP50"] >= FP8_MAP50_TOL, mp
    return out


def test_vit_b_fp8_vs_reference_golden_and_emulation(golden_dir):
    out = _fp8_vs_golden("vit_b", golden_dir, 2)
    # the CPU emulation of the same arithmetic (e4m3 operands with the packer's per-channel weight scales, bf16
    # attention, fp16 stem / neck): the kernels must be its faithful implementation, tile 0
    m, _ = _model("vit_b", "fp8")
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cfg = O.OracleCfg.from_model_type("vit_b", rnd=O.fp16_round)
    cfg.block_fp8 = True
    ref = O.model_forward(torch.from_numpy(synth.make_batch(0, 1)), sd, cfg)
    err = G.rel_l2(out["pred_logits"][:1], ref["pred_logits"])
    print(f"[vit_b/fp8] GPU vs CPU emulation of the fp8 arithmetic: logits={err:.2e}")
    assert err < 5e-3, err


def test_fp8_stream_planes_against_fp32_stream():
    """fp8 mode keeps the blocks' residual stream as bf16 / fp16 planes of rows with a position-wise LayerNorm on the upper plane
    (gemm8.h PLANES; WM_FP8_ROWS=0 = the fp32 stream and the column-tiled LayerNorm kernel).  The two forms carry the same values to
    2^-19 and differ only where bf16(x) in front of the LayerNorm moves an e4m3 rounding -- but e4m3 rounding noise is chaotic: a few
    moved roundings move others downstream, so per element the two streams drift apart to a fraction of their own distance to the
    reference (tools/fp8_tap_check.py, ViT-B: block taps 1.0e-2 .. 4.5e-2 of the reference fixture in BOTH forms, logits 8.7e-3 /
    8.8e-3), while the logits, which average the activation noise over 4096 tokens and share the weights' quantisation error, stay
    together to < 1e-3.  Asserted: taps (read through the planes' merge) within 4e-2 of each other, logits within 3e-3."""
    from wildlifemapper_amd import _native as Nn
    m, _ = _model("vit_b", "fp16")
    hub = m._hub
    depth = synth.MODEL_DIMS["vit_b"].depth
    x = torch.from_numpy(synth.make_batch(0, 2)).to(G.dev())
    ts = torch.tensor([[1024, 1024]] * 2)
    got = {}
    try:
        for rows in ("0", "1"):
            os.environ["WM_FP8_ROWS"] = rows
            hub.set_precision("fp16")
            hub.set_precision("fp8")                          # a fresh native handle reads the switch
            hub.handle(x.device, 2)
            hfc = m.fft(x)
            taps = []
            for which in (depth // 2, depth - 1):
                hub.set_tap(which)
                m.image_encoder(x, hfc)
                taps.append(hub.read_tap(2).cpu())
            hub.set_tap(-2)
            Nn.gemm_variant_counts(reset=True)
            out = m.detect(NestedTensor(x, None), ts)
            torch.cuda.synchronize()
            var = Nn.gemm_variant_counts()
            assert (var.get("fp8_256_planes", 0) == 2 * depth) == (rows == "1"), var
            got[rows] = taps + [out["pred_logits"].cpu()]
    finally:
        os.environ.pop("WM_FP8_ROWS", None)
        hub.set_precision("fp16")
    names = ("mid-block tap", "last-block tap", "logits")
    errs = {n: G.rel_l2(a, b) for n, a, b in zip(names, got["1"], got["0"])}
    print(f"[vit_b/fp8] planes vs fp32 stream: {errs}")
    assert all(torch.isfinite(t).all() for t in got["1"])
    assert errs["mid-block tap"] < 4e-2 and errs["last-block tap"] < 4e-2 and errs["logits"] < 3e-3, errs


def test_fp8_planes_with_bf16_head_and_tail_blocks(golden_dir):
    """The plane form of the fp8 blocks' stream next to blocks of another operand type (WM_FP8_BF16_HEAD / TAIL: the first / last K
    blocks run bf16 on the fp32 stream): fp32 -> planes in front of the first fp8 block, planes -> fp32 behind the last, the neck fed
    from fp32.  Same bounds as the all-fp8 run; the instance counters show 2 plane GEMMs per fp8 block and none for the others."""
    m, _ = _model("vit_b", "fp16")
    try:
        os.environ["WM_FP8_BF16_HEAD"], os.environ["WM_FP8_BF16_TAIL"] = "1", "2"
        m._hub.set_precision("fp16")                          # the next fp8 handle reads the switches
        _fp8_vs_golden("vit_b", golden_dir, 2)
    finally:
        os.environ.pop("WM_FP8_BF16_HEAD", None)
        os.environ.pop("WM_FP8_BF16_TAIL", None)
        m._hub.set_precision("fp16")


def test_fp8_gemm_mask_and_saturation_census(golden_dir):
    """wm_config.fp8_gemms (round 3): ViT-B with only the MLP pair on the fp8 MFMA (qkv / proj / attention bf16).  The GEMM
    instance counters show the mix; its logits error lies between bf16's and all-fp8's.  Then the saturation census
    (wm_debug_saturation_*): zero on the synthetic weights in fp16 mode; a lin1 bias pushed past fp16's range makes the GELU
    hidden clamp at 65504 and the census reports it (the signal for switching such a checkpoint to bf16)."""
    from wildlifemapper_amd import _native as Nn
    fx = np.load(os.path.join(golden_dir, "e2e_vit_b.npz"))
    m, _ = _model("vit_b", "fp8")
    hub = m._hub
    x = torch.from_numpy(synth.make_batch(0, 2)).to(G.dev())
    ts = torch.tensor([[1024, 1024]] * 2)
    errs = {}
    try:
        for name, mask in (("all", Nn.FP8_ALL), ("mlp", Nn.FP8_MLP), ("qkv+proj", Nn.FP8_QKV | Nn.FP8_PROJ)):
            hub.set_fp8_gemms(mask)
            m.detect(NestedTensor(x, None), ts)
            Nn.gemm_variant_counts(reset=True)
            out = m.detect(NestedTensor(x, None), ts)
            torch.cuda.synchronize()
            var = {k: v for k, v in Nn.gemm_variant_counts().items() if v}
            n8 = {"all": 4, "mlp": 2, "qkv+proj": 2}[name] * 12
            assert var.get("fp8_256", 0) + var.get("fp8_256_planes", 0) == n8, (name, var)
            assert (var.get("fp8_256_planes", 0) > 0) == (name == "all"), (name, var)   # planes only where all four GEMMs of a block take e4m3
            lg = out["pred_logits"].cpu().numpy()
            errs[name] = float(np.linalg.norm(lg - fx["pred_logits"]) / np.linalg.norm(fx["pred_logits"]))
    finally:
        hub.set_fp8_gemms(0)
    print(f"[vit_b/fp8 masks] logits rel-L2 vs reference: {errs}")
    assert errs["mlp"] < errs["all"] and errs["qkv+proj"] < errs["all"] and errs["all"] < FP8_LOGIT_TOL
    # --- saturation census, fp16 mode
    m, _ = _model("vit_b", "fp16")
    hub = m._hub
    m.detect(NestedTensor(x, None), ts)
    hub.saturation_enable(True)
    try:
        hub.saturation_read(reset=True)
        m.detect(NestedTensor(x, None), ts)
        clean = hub.saturation_read(reset=True)
        assert set(clean) == set(Nn.SAT_NAMES) and all(v == 0 for v in clean.values()), clean
        bias = m.image_encoder.blocks[5].mlp.lin1.bias
        keep = bias.detach().clone()
        with torch.no_grad():
            bias[:7] += 1.0e5                          # GELU(1e5) = 1e5 > 65504: seven hidden channels clamp on every token
        m.detect(NestedTensor(x, None), ts)
        hot = hub.saturation_read(reset=True)
        with torch.no_grad():
            bias.copy_(keep)
        assert hot["mlp_hidden"] >= 7 * 2 * 4096, hot
        m.detect(NestedTensor(x, None), ts)
        assert all(v == 0 for v in hub.saturation_read().values())
    finally:
        hub.saturation_enable(False)


def test_overflow_words_are_loud():
    """The two always-on range watches (wm_stream_overflow): a residual stream that reaches the fp16 clamp (fp16 mode, folded
    LayerNorm: its operand is the stream itself) and a decoder GEMM operand outside fp16's range (the decoder's GEMMs split fp32
    values into fp16 pairs).  Clean weights raise neither; each perturbation raises its own bit and the drop-in warns at the next
    call; the remedy named in the decoder's warning (WM_GEMM32_F32=1) is a process switch; with the weights restored the outputs are
    bit for bit the clean ones again."""
    import warnings
    m, _ = _model("vit_b", "fp16")
    hub = m._hub
    x = torch.from_numpy(synth.make_batch(0, 4)).to(G.dev())      # 4 tiles: ViT-B's blocks then run folded (256-row-tile GEMMs)
    ts = torch.tensor([[1024, 1024]] * 4)
    m.detect(NestedTensor(x, None), ts)
    torch.cuda.synchronize()
    hub.stream_overflow(reset=True)
    m.detect(NestedTensor(x, None), ts)
    torch.cuda.synchronize()
    assert hub.stream_overflow(reset=True) == 0
    clean_logits = m.detect(NestedTensor(x, None), ts)["pred_logits"].clone()
    # (1) the stream: a lin2 bias of 1e5 on a few channels
    bias = m.image_encoder.blocks[3].mlp.lin2.bias
    keep = bias.detach().clone()
    with torch.no_grad():
        bias[:5] += 1.0e5
    m.detect(NestedTensor(x, None), ts)
    torch.cuda.synchronize()
    with torch.no_grad():
        bias.copy_(keep)
    assert hub.stream_overflow() & 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.detect(NestedTensor(x, None), ts)
    assert any("fp16 clamp" in str(i.message) for i in w), [str(i.message) for i in w]
    torch.cuda.synchronize()
    assert hub.stream_overflow(reset=True) == 0
    # (2) the decoder: one weight of 5000 (x 2^6 leaves fp16's range)
    wt = m.mask_decoder.transformer.layers[0].self_attn.q_proj.weight       # (not the MLP's lin1: its ReLU turns a nan into 0)
    keepw = wt.detach().clone()
    with torch.no_grad():
        wt[3, 7] = 5000.0
    out = m.detect(NestedTensor(x, None), ts)
    torch.cuda.synchronize()
    with torch.no_grad():
        wt.copy_(keepw)
    assert hub.stream_overflow() & 2
    # the row is inf / nan inside the decoder; what reaches the logits need not be (softmax and ReLU swallow a nan), but it is wrong
    assert G.rel_l2(torch.nan_to_num(out["pred_logits"]), clean_logits) > 0.1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = m.detect(NestedTensor(x, None), ts)
    assert any("WM_GEMM32_F32" in str(i.message) for i in w), [str(i.message) for i in w]
    torch.cuda.synchronize()
    assert hub.stream_overflow(reset=True) == 0 and torch.equal(out["pred_logits"], clean_logits)


def test_vit_h_fp8_batch16_vs_reference_golden(golden_dir):
    """configs[4] literally: ViT-H, batch 16, tile 0 = the golden tile."""
    out = _fp8_vs_golden("vit_h", golden_dir, 16)
    lg = out["pred_logits"]
    assert all(not torch.equal(lg[i], lg[j]) for i in range(16) for j in range(i))
    assert bool(torch.isfinite(lg).all())


@pytest.mark.parametrize("fixture", ["e2e_vit_h_tiles1to4.npz", "e2e_vit_h_smooth.npz", "e2e_vit_h_padded768.npz"])
@pytest.mark.parametrize("prec", ["fp16", "bf16", "fp8"])
def test_vit_h_more_tiles_vs_reference_golden(prec, fixture, golden_dir):
    """The end-to-end tolerance on more ViT-H inputs, each fixture run as one batch: four more noise tiles (1..4) and two
    tiles with low-frequency content (`synth.make_tile_u8(smooth=True)`: pass-band and stop-band energy for the FFT high-pass),
    and the same two with content only in the top-left 768 x 768 and zeros elsewhere, which is what the reference's val pipeline
    feeds the model (resize to 768, normalise, zero-pad to 1024: dataloader_coco.py:288, utils/misc.py:50-64).
    The fixtures hold the reference modules' logits / boxes and the NMS lists (oracle/gen_golden.py --only vit_h_tiles /
    vit_h_smooth).  Per tile: logits within the bar of the precision and, for the 16-bit modes, the NMS index list identical."""
    fx = np.load(os.path.join(golden_dir, fixture))
    n, first = int(fx["n_tiles"]), int(fx["first_tile"])
    m, _ = _model("vit_h", prec)
    x = torch.from_numpy(synth.make_batch(first, n, smooth="smooth" in fx.files))
    if "content" in fx.files:
        c = int(fx["content"])
        x[:, :, c:, :] = 0
        x[:, :, :, c:] = 0
    x = x.to(G.dev())
    with torch.no_grad():
        out = m.detect(x, torch.tensor([[1024, 1024]] * n))
    lg, bx = out["pred_logits"].cpu().numpy(), out["pred_boxes"].cpu().numpy()
    rec = split_records(out["records"].cpu())
    errs, same = [], []
    for t in range(n):
        errs.append(float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])))
        same.append(_nms_positions(rec, t) == fx[f"pp{t}_nms_index"].tolist())
    berr = float(np.abs(bx - fx["pred_boxes"]).max())
    print(f"[vit_h/{prec}] {fixture} tiles {first}..{first + n - 1}: logits per tile " + " ".join(f"{e:.2e}" for e in errs) + f" boxes_maxabs={berr:.2e} NMS identical: {same}")
    if prec == "fp8":
        assert max(errs) < FP8_LOGIT_TOL and berr < FP8_BOX_TOL, (errs, berr)
    else:
        assert max(errs) < LOGIT_TOL[prec], errs            # north_star's 1e-3 on every tile
        assert berr < 5 * LOGIT_TOL[prec], berr
        assert all(same), same


def test_vit_h_second_weight_seed_vs_reference_golden(golden_dir):
    """A DIFFERENT set of synthetic weights (synth seed 1; tests/golden/e2e_vit_h_seed1.npz: the reference modules on tiles 0
    and 1, oracle/gen_golden.py --only vit_h_seed1).  fp16 operands (the default mode) meet north_star's 1e-3 here as on seed 0
    (measured 1.7-1.8e-4).  bf16 operands do NOT: 2.0e-3 (seed 0: 7.2-8.2e-4); what bf16 reaches depends on the weights, which is
    why it is not the default.  Its bound here is what was measured + 25 %.  NMS lists identical in both.
    The resident model's parameters are replaced in place (the hub re-packs them) and restored afterwards."""
    fx = np.load(os.path.join(golden_dir, "e2e_vit_h_seed1.npz"))
    n, first, seed = int(fx["n_tiles"]), int(fx["first_tile"]), int(fx["weight_seed"])
    m, _ = _model("vit_h", "bf16")
    base = {k: v.detach().clone() for k, v in m.state_dict().items()}
    try:
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h", seed).items()}, strict=True)
        x = torch.from_numpy(synth.make_batch(first, n)).to(G.dev())
        for prec in ("fp16", "bf16", "fp8"):
            m._hub.set_precision(prec)
            with torch.no_grad():
                out = m.detect(x, torch.tensor([[1024, 1024]] * n))
            lg = out["pred_logits"].cpu().numpy()
            rec = split_records(out["records"].cpu())
            errs = [float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])) for t in range(n)]
            same = [_nms_positions(rec, t) == fx[f"pp{t}_nms_index"].tolist() for t in range(n)]
            print(f"[vit_h/{prec}/seed {seed}] logits per tile " + " ".join(f"{e:.2e}" for e in errs) + f" NMS identical: {same}")
            assert max(errs) < {"fp16": LOGIT_TOL["fp16"], "bf16": 2.5e-3, "fp8": FP8_LOGIT_TOL}[prec], errs     # fp8 measured 2.5-2.6e-2
            assert prec == "fp8" or all(same), same
    finally:
        m.load_state_dict(base, strict=True)


def test_vit_h_outlier_weight_profile_vs_reference_golden(golden_dir):
    """Activation outliers: synth profile "outlier" -- in blocks 5, 13, 21, 29 six LayerNorm gammas x50 in
    norm1 / norm2, eight lin1 rows x30, two lin2 output channels x40 ("massive" channels that stay in the residual stream) --
    against tests/golden/e2e_vit_h_outlier.npz (the reference's own modules with those weights on tiles 0, 1; oracle/gen_golden.py
    --only vit_h_outlier).  fp16 operands, the default, must still meet north_star's 1e-3 on the logits with identical NMS
    lists, and the saturation census must find no operand value at fp16's clamp; bf16's number is printed and bounded loosely."""
    fx = np.load(os.path.join(golden_dir, "e2e_vit_h_outlier.npz"))
    n, first = int(fx["n_tiles"]), int(fx["first_tile"])
    assert str(fx["profile"]) == "outlier"
    peak = fx["resid_max_rms_tile0"]
    assert peak[:, 0].max() > 20 * peak[0, 1]              # the fixture's residual stream does carry outliers (max |x| vs block-0 rms)
    m, _ = _model("vit_h", "fp16")
    base = {k: v.detach().clone() for k, v in m.state_dict().items()}
    try:
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h", 0, profile="outlier").items()}, strict=True)
        x = torch.from_numpy(synth.make_batch(first, n)).to(G.dev())
        for prec in ("fp16", "bf16"):
            m._hub.set_precision(prec)
            with torch.no_grad():
                m.detect(x, torch.tensor([[1024, 1024]] * n))
                m._hub.saturation_enable(True)
                m._hub.saturation_read(reset=True)
                out = m.detect(x, torch.tensor([[1024, 1024]] * n))
                sat = m._hub.saturation_read(reset=True)
                m._hub.saturation_enable(False)
            lg = out["pred_logits"].cpu().numpy()
            rec = split_records(out["records"].cpu())
            errs = [float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])) for t in range(n)]
            same = [_nms_positions(rec, t) == fx[f"pp{t}_nms_index"].tolist() for t in range(n)]
            print(f"[vit_h/{prec}/outlier profile] logits per tile " + " ".join(f"{e:.2e}" for e in errs) + f" NMS identical: {same}  clamped operands: {sat}")
            assert all(v == 0 for v in sat.values()), sat
            assert max(errs) < {"fp16": LOGIT_TOL["fp16"], "bf16": 5e-3}[prec], errs
            assert prec != "fp16" or all(same), same
    finally:
        m.load_state_dict(base, strict=True)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_vit_h_folded_layernorm_vs_reference_golden(prec, golden_dir):
    """WM_CFG_FOLD_LN: the blocks' LayerNorms folded into the GEMMs around them (statistics from the residual GEMM's epilogue,
    normalisation in the qkv / lin1 epilogue).  Against the reference fixtures: tile 0 (with nine block taps' worth of logits
    checked through the final outputs), tiles 1..4 and the outlier weight profile; per tile the logits bar of the precision
    and, in fp16 mode, identical NMS lists.  And batch invariance: tile 0 alone (B = 1: statistics from the standalone kernel,
    half-width residual GEMMs), in a batch of 4 and in a batch of 16 (statistics from the producing GEMMs) gives the same bits."""
    from wildlifemapper_amd import _native as Nn
    m, _ = _model("vit_h", prec)
    hub = m._hub
    fold_was = hub.fold_ln
    hub.close(); hub.fold_ln = "all"                      # bf16-operand blocks fold only on request (WM_CFG_FOLD_LN_BF16)
    base = None
    try:
        outs = {}
        for B in (1, 4, 16):
            x = torch.from_numpy(synth.make_batch(0, B)).to(G.dev())
            with torch.no_grad():
                m.detect(x, torch.tensor([[1024, 1024]] * B))
                Nn.gemm_variant_counts(reset=True)
                outs[B] = {k: v.cpu() for k, v in m.detect(x, torch.tensor([[1024, 1024]] * B)).items()}
            var = {k: v for k, v in Nn.gemm_variant_counts().items() if v}
            if B >= 4:
                # producers: 32 x proj + 32 x lin2 on the split-stream instance + the stem's proj_back (fp32 residual in, planes
                # out) where its fp16 operand type is also block 0's (fp16 mode)
                assert var.get("v5_320_split", 0) == 64 and var.get("v5_320_foldp", 0) == (1 if prec == "fp16" else 0), var
            else:
                assert var.get("v5_320_foldp", 0) == 0 and var.get("v5_320_split", 0) == 0, var
        for B in (4, 16):
            assert torch.equal(outs[B]["pred_logits"][0], outs[1]["pred_logits"][0]), B
            assert torch.equal(outs[B]["pred_boxes"][0], outs[1]["pred_boxes"][0]), B
        for fixture in ("e2e_vit_h.npz", "e2e_vit_h_tiles1to4.npz"):
            fx = np.load(os.path.join(golden_dir, fixture))
            n, first = int(fx["n_tiles"]), int(fx["first_tile"])
            lg = outs[16]["pred_logits"][first:first + n].numpy()
            rec = split_records(outs[16]["records"][first:first + n])
            errs = [float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])) for t in range(n)]
            same = [_nms_positions(rec, t) == fx[f"pp{t}_nms_index"].tolist() for t in range(n)]
            print(f"[vit_h/{prec}/folded LN] {fixture}: logits per tile " + " ".join(f"{e:.2e}" for e in errs) + f" NMS identical: {same}")
            # bf16 folded is opt-in: its logits error is another draw of bf16's ~1e-3 (measured 1.29-1.36e-3 on these tiles against
            # 7.5-8.0e-4 with the LayerNorm kernel; ViT-L the other way round: include/wm_hip.h WM_CFG_FOLD_LN), bound = measured + 10 %
            assert max(errs) < (LOGIT_TOL[prec] if prec == "fp16" else 1.5e-3), errs
            assert all(same), same
        # the outlier weight profile through the folded path
        fx = np.load(os.path.join(golden_dir, "e2e_vit_h_outlier.npz"))
        base = {k: v.detach().clone() for k, v in m.state_dict().items()}
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h", 0, profile="outlier").items()}, strict=True)
        x = torch.from_numpy(synth.make_batch(0, 2)).to(G.dev())
        with torch.no_grad():
            out = m.detect(x, torch.tensor([[1024, 1024]] * 2))
        lg = out["pred_logits"].cpu().numpy()
        rec = split_records(out["records"].cpu())
        errs = [float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])) for t in range(2)]
        same = [_nms_positions(rec, t) == fx[f"pp{t}_nms_index"].tolist() for t in range(2)]
        print(f"[vit_h/{prec}/folded LN/outlier profile] logits per tile " + " ".join(f"{e:.2e}" for e in errs) + f" NMS identical: {same}")
        assert max(errs) < {"fp16": LOGIT_TOL["fp16"], "bf16": 5e-3}[prec], errs
        assert prec != "fp16" or all(same), same
    finally:
        if base is not None:
            m.load_state_dict(base, strict=True)
        hub.close(); hub.fold_ln = fold_was


def test_input_pipeline_resize_bit_exact_vs_pil(golden_dir):
    """N1 with the val transform's resize (dataloader_coco.py:288): uint8 frame -> PIL-bilinear resample -> ToTensor ->
    Normalize -> zero-padded 1024^2 tile, on the GPU.  Integer work: bit-exact against (a) vectors PIL itself produced
    (tests/golden/resize_pil.npz) and (b) the oracle's restatement on a full-size 3648 x 5472 frame (val.json's frame size,
    resized to 512 x 768 as the reference would), then through the bit-exact normalise twin."""
    from oracle import pil_resize as R
    from wildlifemapper_amd.preprocess import tiles_from_u8, resized_size
    fx = np.load(os.path.join(golden_dir, "resize_pil.npz"))
    for i, (h, w, size, mx) in enumerate(fx["cases"]):
        img, want_u8 = fx[f"in{i}"], fx[f"out{i}"]
        oh, ow = want_u8.shape[:2]
        assert resized_size(int(h), int(w), int(size), int(mx)) == (oh, ow)
        got = tiles_from_u8(torch.from_numpy(img[None]).to(G.dev()), resize=(int(size), int(mx))).cpu().numpy()[0]
        want = synth.normalize_tile(want_u8)
        assert np.array_equal(got[:, :oh, :ow], want), i
        assert not got[:, oh:, :].any() and not got[:, :, ow:].any()
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (2, 3648, 5472, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:3648, 0:5472]
    frame[1, ..., 0] = (yy // 16 + xx // 16) % 256                  # structured content in one channel
    got = tiles_from_u8(torch.from_numpy(frame).to(G.dev()), resize=(768, 768)).cpu().numpy()
    for b in range(2):
        ref_u8 = R.val_transform_u8(frame[b], 768, 768)
        assert ref_u8.shape == (512, 768, 3)
        assert np.array_equal(got[b][:, :512, :768], synth.normalize_tile(ref_u8)), b
        assert not got[b][:, 512:, :].any() and not got[b][:, :, 768:].any()
    with pytest.raises(RuntimeError, match="canvas"):
        tiles_from_u8(torch.zeros(1, 2000, 2000, 3, dtype=torch.uint8, device=G.dev()), resize=(1100, 0))


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """The N > 1 path of bench.py (self-launch through torch.distributed.run, tile sharding, the per-step all-gather of box
    records, max-over-ranks timing, one JSON line on rank 0) rehearsed with two ranks sharing this box's single GPU.  The
    collective runs on gloo here (RCCL refuses two ranks on one device); on a multi-GPU node the driver launches the same
    script with the nccl backend.  Three GPU processes in total (this one + 2 ranks), below the box's process guard."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--model", "vit_b",
           "--batch", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["tiles_per_step"] == 4 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["steps"] == 2 and line["warmup"] == 1
    assert "all-gather of box records" in line["config"]["parallelism"]
    assert line["roofline"]["achieved"] > 0 and line["cpu_baseline"] is None
