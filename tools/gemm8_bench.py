#!/usr/bin/env python3
"""Dev tool: time the fp8 GEMM (gemm8.h) on the block shapes, random data."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--shapes", default="qkv,proj,lin1,lin2")
a = ap.parse_args()
M = a.batch * 4096
shapes = {"qkv": (M, 3840, 1280, 0, "16"), "proj": (M, 1280, 1280, 0, "f32"), "lin1": (M, 5120, 1280, 1, "8"), "lin1_noact": (M, 5120, 1280, 0, "8"), "lin2": (M, 1280, 5120, 0, "f32")}
dev = G.dev()
for name in a.shapes.split(","):
    m, n, k, act, mode = shapes[name]
    a8 = G.to_fp8(torch.randn(m, k, device=dev))
    w8, sc = G.quant_weight_fp8(torch.randn(n, k, device=dev) / math.sqrt(k))
    bias = torch.randn(n, device=dev)
    res = torch.randn(m, n, device=dev) if mode == "f32" else None
    from wildlifemapper_amd import _native as N
    o16 = torch.empty((m, n), device=dev, dtype=torch.bfloat16) if mode == "16" else None
    o8 = torch.empty((m, n), device=dev, dtype=torch.uint8) if mode == "8" else None
    def run():
        N.check(N.lib().wm_op_gemm8(N.ptr(a8), N.ptr(w8), N.ptr(sc), N.ptr(bias), N.ptr(res), N.ptr(res), N.ptr(o16), N.ptr(o8), m, n, k, act, 0, G.sp()))
    best = []
    for r in range(a.rounds):
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): run()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / a.iters)
    us = sorted(best)[len(best) // 2]
    print(f"{name:5s} M={m} N={n} K={k} out={mode}: {us:8.1f} us  {2.0*m*n*k/us/1e6:8.1f} TFLOP/s", flush=True)
