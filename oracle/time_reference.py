#!/usr/bin/env python3
"""Time the reference's own modules (imported in place, as oracle/gen_golden.py does) on this container's CPU:
BASELINE.md section 3 (i).  ViT-H or ViT-B, one synthetic tile, fp32, torch.no_grad, 1 warm-up + N timed forwards.
Only runs where /root/reference exists (never on the GPU box).  usage: python oracle/time_reference.py vit_h [n]"""
import os, statistics, sys, time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as GG  # noqa: E402
from oracle import wm_oracle as O  # noqa: E402
from wildlifemapper_amd import synth  # noqa: E402

mt = sys.argv[1] if len(sys.argv) > 1 else "vit_h"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
enc, dec, pe = GG.build_ref_model(mt)
GG.load_synth(enc, "image_encoder."); GG.load_synth(dec, "mask_decoder."); GG.load_synth(pe, "prompt_encoder.")
x = torch.from_numpy(synth.make_batch(0, 1))
hfc = O.hfc_fft(x)
times = []
with torch.no_grad():
    for i in range(n + 1):
        t0 = time.time()
        emb = enc(x, hfc)
        t1 = time.time()
        dec(image_embeddings=emb, image_pe=pe.get_dense_pe(), sparse_prompt_embeddings=None, dense_prompt_embeddings=None,
            multimask_output=False, hfc_embed=None)
        t2 = time.time()
        print(f"{mt} forward {i}: encoder {t1 - t0:.1f} s, decoder {t2 - t1:.2f} s", flush=True)
        if i > 0:
            times.append(t2 - t0)
print(f"{mt}: median {statistics.median(times):.1f} s/tile = {1 / statistics.median(times):.4f} tiles/s, "
      f"{torch.get_num_threads()} threads, {os.cpu_count()} cpus")
