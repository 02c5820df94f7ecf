#!/usr/bin/env python3
"""Dev tool (round 4): ViT-H logits error against the reference fixtures, nine seed-0 tiles + two seed-1 tiles, for the ways the
blocks' LayerNorm can run in a precision mode: its own kernel (WM_LN_FOLD=0), folded with gamma (.) W rounded twice (WM_FOLD_FROM16=1,
round 3) or once (default), fp32 stream (WM_STREAM_SPLIT=0) or split stream (default).  usage: bf16_fold_study.py [bf16] [fp16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wildlifemapper_amd import synth
from wildlifemapper_amd.engine import split_records
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM
dev = torch.device("cuda", 0)
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sam, _, _ = sam_model_registry["vit_h"](None, None)
m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
MODES = [("LN kernel", {"WM_LN_FOLD": "0"}), ("fold16 fp32-stream", {"WM_FOLD_FROM16": "1", "WM_STREAM_SPLIT": "0"}),
         ("fold32 fp32-stream", {"WM_STREAM_SPLIT": "0"}), ("fold32 split", {})]
def nms_pos(rec, b):
    flags, rank = rec["flags"][b], rec["nms_rank"][b]
    pos = torch.cumsum(((flags & 2) != 0).long(), 0) - 1
    slots = torch.nonzero((flags & 4) != 0).flatten()
    return pos[slots[torch.argsort(rank[slots])]].tolist()
for seed, fixtures in ((0, ("e2e_vit_h.npz", "e2e_vit_h_tiles1to4.npz", "e2e_vit_h_smooth.npz", "e2e_vit_h_padded768.npz")), (1, ("e2e_vit_h_seed1.npz",))):
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h", seed).items()}, strict=True)
    for prec in sys.argv[1:] or ["bf16"]:
        for name, env in MODES:
            for k in ("WM_LN_FOLD", "WM_FOLD_FROM16", "WM_STREAM_SPLIT"):
                os.environ.pop(k, None)
            os.environ.update(env)
            m._hub.fold_ln = os.environ.get("WM_LN_FOLD", "1") != "0"
            m._hub.close()
            m._hub.set_precision(prec)
            errs, same = [], []
            for fxn in fixtures:
                fx = np.load(os.path.join(G, fxn))
                n, first = int(fx["n_tiles"]), int(fx["first_tile"])
                x = torch.from_numpy(synth.make_batch(first, n, smooth="smooth" in fx.files))
                if "content" in fx.files:
                    c = int(fx["content"])
                    x[:, :, c:, :] = 0
                    x[:, :, :, c:] = 0
                x = x.to(dev)
                with torch.no_grad():
                    out = m.detect(x, torch.tensor([[1024, 1024]] * n))
                lg = out["pred_logits"].cpu().numpy()
                rec = split_records(out["records"].cpu())
                errs += [float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])) for t in range(n)]
                same += [nms_pos(rec, t) == fx[f"pp{t}_nms_index"].tolist() for t in range(n)]
            print(f"seed {seed} {prec} {name:20s}: " + " ".join(f"{e:.2e}" for e in errs) + f"  rms {np.sqrt(np.mean(np.square(errs))):.2e} max {max(errs):.2e}  NMS {sum(same)}/{len(same)}", flush=True)
