#!/usr/bin/env python3
"""Dev tool: ViT-H logits error against the reference fixtures for weight seeds 0 and 1, per precision mode (and WM_FP16_TAIL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wildlifemapper_amd import synth
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM
dev = torch.device("cuda", 0)
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sam, _, _ = sam_model_registry["vit_h"](None, None)
m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
for seed, fxn in ((0, "e2e_vit_h_tiles1to4.npz"), (1, "e2e_vit_h_seed1.npz")):
    fx = np.load(os.path.join(G, fxn))
    n, first = int(fx["n_tiles"]), int(fx["first_tile"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h", seed).items()}, strict=True)
    x = torch.from_numpy(synth.make_batch(first, n)).to(dev)
    for prec in sys.argv[1:] or ["bf16", "fp16"]:
        m._hub.set_precision(prec)
        with torch.no_grad():
            lg = m.detect(x, torch.tensor([[1024, 1024]] * n))["pred_logits"].cpu().numpy()
        errs = [float(np.linalg.norm(lg[t] - fx["pred_logits"][t]) / np.linalg.norm(fx["pred_logits"][t])) for t in range(n)]
        print(f"seed {seed} {prec} WM_FP16_TAIL={os.environ.get('WM_FP16_TAIL', '0')}: " + " ".join(f"{e:.2e}" for e in errs), flush=True)
