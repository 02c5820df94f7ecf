#!/usr/bin/env python3
"""Dev tool (round 4): N1 input pipeline (wm_preprocess_u8_resized) on resident 3648 x 5472 uint8 frames: us per frame and the
algorithmic GB/s (frame read once + tile written once), streaming kernels vs the generic ones (WM_RESIZE_GENERIC=1), bit-compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wildlifemapper_amd import preprocess
dev = torch.device("cuda", 0)
torch.manual_seed(1)
for B in (16, 1):
    frames = torch.randint(0, 256, (B, 3648, 5472, 3), dtype=torch.uint8, device=dev)
    outs = {}
    for mode in ("streaming", "generic"):
        os.environ["WM_RESIZE_GENERIC"] = "1" if mode == "generic" else "0"
        for _ in range(3):
            out = preprocess.tiles_from_u8(frames, resize=(768, 768))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = preprocess.tiles_from_u8(frames, resize=(768, 768))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        outs[mode] = out
        by = 3648 * 5472 * 3 + 3 * 1024 * 1024 * 4
        print(f"B={B:2d} {mode:10s}: {dt / B * 1e6:8.1f} us per frame, {by * B / dt / 1e9:7.1f} GB/s algorithmic = {by * B / dt / 8e12:.3f} of 8 TB/s", flush=True)
    print("   bit-identical:", torch.equal(outs["streaming"], outs["generic"]), flush=True)
