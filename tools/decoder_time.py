#!/usr/bin/env python3
"""Dev tool: wall time of wm_decoder_forward alone (ViT-H handle, B tiles, resident embedding), HIP events around N calls."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import gpu_util as G
from wildlifemapper_amd import synth
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mt = "vit_b"                                   # the decoder is the same for every encoder size
sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(mt).items()}
sam, crit, post = sam_model_registry[mt](None, None)
m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
m.load_state_dict(sd, strict=True)
hub = m._hub
hub.set_precision("fp16")
emb = torch.randn(B, 256, 64, 64, device=G.dev())
hub.handle(emb.device, B)
for _ in range(5): hub.decoder_forward(emb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): hub.decoder_forward(emb)
e1.record(); torch.cuda.synchronize()
print(f"decoder_forward B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (WM_GEMM32_F32={os.environ.get('WM_GEMM32_F32', '0')})")
