#!/usr/bin/env python3
"""Dev tool: time the encoder attention kernels on the ViT-H shapes (random data)."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--prec", default="bf16")
a = ap.parse_args()
dev = G.dev()
B, heads, hd = a.batch, 16, 80
D = heads * hd
qkv = G.to16(torch.randn(B * 4096, 3 * D, device=dev) * 0.5, a.prec)
bias = torch.randn(3 * D, device=dev) * 0.1

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / a.iters

for name, window in (("global", 0), ("window", 14)):
    S = 64 if window == 0 else 14
    rel_h = torch.randn(2 * S - 1, hd, device=dev) * 0.1
    rel_w = torch.randn(2 * S - 1, hd, device=dev) * 0.1
    us = timeit(lambda: G.encoder_attention(qkv, bias, rel_h, rel_w, B, heads, hd, window, a.prec))
    nk = 4096 if window == 0 else 196
    fl = 4.0 * B * heads * 4096 * nk * hd
    print(f"{name:7s} B={B} heads={heads} hd={hd}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s (useful)", flush=True)
# HFC cross attention: 8 heads x 128, no rel-pos
q = G.to16(torch.randn(B * 4096, 1024, device=dev) * 0.5, "fp16"); kv = G.to16(torch.randn(B * 4096, 2048, device=dev) * 0.5, "fp16")
us = timeit(lambda: G.mha16(q, kv[:, :1024], kv[:, 1024:], B, 8, 128, 4096, 4096, "fp16"))
print(f"hfc     B={B} heads=8 hd=128: {us:8.1f} us  {4.0 * B * 8 * 4096 * 4096 * 128 / us / 1e6:7.1f} TFLOP/s", flush=True)
