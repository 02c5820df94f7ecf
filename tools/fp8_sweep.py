#!/usr/bin/env python3
"""Dev tool (round 3): which GEMMs of a block run on the fp8 MFMA (wm_config.fp8_gemms: qkv / proj / MLP pair)
x how many head / tail blocks stay bf16 (WM_FP8_BF16_HEAD / _TAIL), ViT-H: logits error, NMS-list identity and mAP against the
reference fixtures (11 tiles with weight seed 0: tile 0, tiles 1..4, 2 smooth, 2 padded-768; optionally seed 1), and tiles/s at B = 16."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import wm_oracle as O
from wildlifemapper_amd import _native as N, synth
from wildlifemapper_amd.coco_eval import map_vs_reference
from wildlifemapper_amd.engine import split_records
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM

ap = argparse.ArgumentParser()
ap.add_argument("--masks", default="7,4,5,6,3,1,2")
ap.add_argument("--headtail", default="0:0,0:8,8:8")
ap.add_argument("--seed1", action="store_true")
ap.add_argument("--out", default="gpurun_out/fp8_sweep.json")
a = ap.parse_args()
dev = torch.device("cuda:0")
gold = os.path.join(ROOT, "tests", "golden")


def nms_positions(rec, b):
    flags, rank = rec["flags"][b], rec["nms_rank"][b]
    pos = torch.cumsum(((flags & 2) != 0).long(), 0) - 1
    slots = torch.nonzero((flags & 4) != 0).flatten()
    return pos[slots[torch.argsort(rank[slots])]].tolist()


def dets(rec, b):
    kept = (rec["flags"][b] & 4) != 0
    order = torch.argsort(rec["nms_rank"][b][kept])
    return {"boxes": rec["boxes"][b][kept][order].numpy(), "scores": rec["scores"][b][kept][order].numpy(), "labels": rec["labels"][b][kept][order].numpy()}


def fixtures(seed):
    """[(input batch, logits, boxes, nms lists)]"""
    out = []
    names = ["e2e_vit_h.npz", "e2e_vit_h_tiles1to4.npz", "e2e_vit_h_smooth.npz", "e2e_vit_h_padded768.npz"] if seed == 0 else ["e2e_vit_h_seed1.npz"]
    for name in names:
        fx = np.load(os.path.join(gold, name))
        n, first = int(fx["n_tiles"]), int(fx["first_tile"])
        x = torch.from_numpy(synth.make_batch(first, n, smooth="smooth" in fx.files))
        if "content" in fx.files:
            c = int(fx["content"]); x[:, :, c:, :] = 0; x[:, :, :, c:] = 0
        out.append((x, fx["pred_logits"], fx["pred_boxes"], [fx[f"pp{t}_nms_index"].tolist() for t in range(n)]))
    return out


sam, _, _ = sam_model_registry["vit_h"](None, None)
m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
results = []
xb = torch.from_numpy(synth.make_batch(0, 16)).to(dev)
tsb = torch.full((16, 2), 1024.0, device=dev)
for seed in ([0, 1] if a.seed1 else [0]):
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h", seed).items()}, strict=True)
    fxs = fixtures(seed)
    for ht in a.headtail.split(","):
        head, tail = (int(v) for v in ht.split(":"))
        os.environ["WM_FP8_BF16_HEAD"], os.environ["WM_FP8_BF16_TAIL"] = str(head), str(tail)
        for mask in (int(v) for v in a.masks.split(",")):
            hub = m._hub
            hub.close()
            hub.set_precision("fp8"); hub.fp8_gemms = mask
            errs, same, pred, gt, k = [], [], {}, {}, 0
            with torch.no_grad():
                for x, lg_ref, bx_ref, nms in fxs:
                    n = x.shape[0]
                    out = m.detect(x.to(dev), torch.tensor([[1024, 1024]] * n))
                    lg = out["pred_logits"].cpu().numpy()
                    rec = split_records(out["records"].cpu())
                    for t in range(n):
                        errs.append(float(np.linalg.norm(lg[t] - lg_ref[t]) / np.linalg.norm(lg_ref[t])))
                        same.append(nms_positions(rec, t) == nms[t])
                        pred[k] = dets(rec, t)
                        d = O.detect(O.postprocess(torch.from_numpy(lg_ref[t][None]), torch.from_numpy(bx_ref[t][None]), torch.tensor([[1024, 1024]]))[0])
                        gt[k] = {"boxes": d["boxes"].numpy(), "scores": d["scores"].numpy(), "labels": d["labels"].numpy()}
                        k += 1
                mp = map_vs_reference(pred, gt)
                tps = None
                if seed == 0:
                    for _ in range(2): m.detect(xb, tsb)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(5): m.detect(xb, tsb)
                    torch.cuda.synchronize(); tps = 16 * 5 / (time.perf_counter() - t0)
            r = {"seed": seed, "fp8_gemms": mask, "bf16_head": head, "bf16_tail": tail, "tiles": len(errs), "logits_max": max(errs), "logits_mean": float(np.mean(errs)),
                 "nms_identical": int(sum(same)), "mAP": mp["mAP"], "mAP50": mp["mAP50"], "tiles_per_s": tps}
            results.append(r)
            names = {1: "qkv", 2: "proj", 4: "mlp"}
            print(f"seed {seed} fp8 GEMMs {'+'.join(v for b_, v in names.items() if mask & b_):13s} bf16 head/tail {head}/{tail}: logits max {r['logits_max']:.2e} mean {r['logits_mean']:.2e}  "
                  f"NMS identical {r['nms_identical']}/{len(same)}  mAP {mp['mAP']:.3f} mAP50 {mp['mAP50']:.3f}" + (f"  {tps:6.1f} tiles/s" if tps else ""), flush=True)
os.makedirs(os.path.dirname(os.path.join(ROOT, a.out)), exist_ok=True)
json.dump(results, open(os.path.join(ROOT, a.out), "w"), indent=1)
