#!/usr/bin/env python3
"""Dev tool: fp8 mode, block taps of ViT-B against the reference fixture, with the stream as planes (default) and as fp32 rows (WM_FP8_ROWS=0)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import gpu_util as G
from wildlifemapper_amd import synth
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM
from wildlifemapper_amd.segment_anything.utils.misc import NestedTensor

mt = "vit_b"
fx = np.load(os.path.join(R, "tests", "golden", f"e2e_{mt}.npz"))
n = int(fx["n_tiles"])
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(mt).items()}
sam, crit, post = sam_model_registry[mt](None, None)
m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
m.load_state_dict(sd, strict=True)
x = torch.from_numpy(synth.make_batch(int(fx["first_tile"]), n)).to(G.dev())
depth = synth.MODEL_DIMS[mt].depth
def sample(t, k):
    flat = t.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, k).long()
    return flat[idx.to(flat.device)].float().cpu().numpy()

import test_gpu_e2e as T
for rows in ("0", "1"):
    os.environ["WM_FP8_ROWS"] = rows
    hub = m._hub
    hub.set_precision("fp16"); hub.set_precision("fp8")
    hub.handle(x.device, n)
    hfc = m.fft(x)
    line = []
    for which in sorted(set(range(0, depth, max(1, depth // 8))) | {depth // 2, depth - 1}):
        hub.set_tap(which)
        m.image_encoder(x, hfc)
        tap = hub.read_tap(n)
        ref = fx[f"block{which}_sample"]
        err = np.linalg.norm(T._sample(tap, 2048) - ref) / np.linalg.norm(ref)
        line.append(f"{which}:{err:.3e}")
    hub.set_tap(-2)
    out = m.detect(NestedTensor(x, None), torch.tensor([[1024, 1024]] * n))
    lg = out["pred_logits"][:n].cpu().numpy()
    print(f"WM_FP8_ROWS={rows}: taps vs reference {' '.join(line)}  logits {np.linalg.norm(lg - fx['pred_logits']) / np.linalg.norm(fx['pred_logits']):.3e}", flush=True)
