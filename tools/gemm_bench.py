#!/usr/bin/env python3
"""Dev tool: time the 16-bit MFMA GEMM on the block shapes (random data), optionally under rocprofv3."""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--shapes", default="qkv,proj,lin1,lin2")
ap.add_argument("--prec", default="bf16")
ap.add_argument("--custom", default="", help="semicolon-separated M,N,K triples")
ap.add_argument("--act", type=int, default=0)
ap.add_argument("--f32out", action="store_true")
ap.add_argument("--residual", action="store_true", help="fp32 residual added in the epilogue (implies --f32out)")
ap.add_argument("--packed", action="store_true", help="A and W in LDS-image order (as the engine feeds the block GEMMs)")
ap.add_argument("--out-packed", action="store_true", help="with --packed: the 16-bit output in LDS-image order too (lin1's form)")
a = ap.parse_args()
M = a.batch * 4096
shapes = {"qkv": (M, 3840, 1280), "proj": (M, 1280, 1280), "lin1": (M, 5120, 1280), "lin2": (M, 1280, 5120),
          "hfc": (M, 1024, 1024), "neck": (M, 256, 2304)}
dev = G.dev()
names = a.shapes.split(",")
if a.custom:
    names = []
    for i, t in enumerate(a.custom.split(";")):
        shapes[f"c{i}"] = tuple(int(v) for v in t.split(","))
        names.append(f"c{i}")
for name in names:
    m, n, k = shapes[name]
    A = G.to16(torch.randn(m, k, device=dev), a.prec)
    W = G.to16(torch.randn(n, k, device=dev) / math.sqrt(k), a.prec)
    bias = torch.randn(n, device=dev)
    res = torch.randn(m, n, device=dev) if a.residual else None
    f32 = a.f32out or a.residual
    layout = 0
    if a.packed:
        from wildlifemapper_amd import _native as Nn
        A, W = G.pack16(A), G.pack16(W)
        layout = Nn.GEMM_W_PACKED | Nn.GEMM_A_PACKED | (Nn.GEMM_OUT_PACKED if a.out_packed else 0)
    for _ in range(3):
        G.gemm16(A, W, bias, residual=res, act=a.act, prec=a.prec, want32=f32, want16=not f32, layout=layout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        G.gemm16(A, W, bias, residual=res, act=a.act, prec=a.prec, want32=f32, want16=not f32, layout=layout)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"{name:5s} M={m} N={n} K={k}: {us:8.1f} us  {2.0*m*n*k/us/1e6:8.1f} TFLOP/s", flush=True)
