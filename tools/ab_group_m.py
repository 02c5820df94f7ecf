#!/usr/bin/env python3
"""Dev tool: A/B of the staggered GEMM's tile-order group size (WM_GEMM_GROUP_M) in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24), block shapes at --batch tiles."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--groups", default="8,1,2,3,4,16")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
M = a.batch * 4096
shapes = {"qkv": (M, 3840, 1280, 0, False), "lin1": (M, 5120, 1280, 1, False), "proj": (M, 1280, 1280, 0, True), "lin2": (M, 1280, 5120, 0, True)}
dev = G.dev()
groups = [int(g) for g in a.groups.split(",")]
for name, (m, n, k, act, res) in shapes.items():
    A = G.to16(torch.randn(m, k, device=dev), "bf16")
    W = G.to16(torch.randn(n, k, device=dev) / math.sqrt(k), "bf16")
    bias = torch.randn(n, device=dev)
    R = torch.randn(m, n, device=dev) if res else None
    times = {g: [] for g in groups}
    def run():
        G.gemm16(A, W, bias, residual=R, act=act, prec="bf16", want32=res, want16=not res)
    for rnd in range(a.rounds):
        for g in groups:
            os.environ["WM_GEMM_GROUP_M"] = str(g)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): run()
            e1.record(); torch.cuda.synchronize()
            times[g].append(e0.elapsed_time(e1) * 1e3 / a.iters)
    base = sorted(times[groups[0]])[len(times[groups[0]]) // 2]
    print(f"{name:5s} M={m} N={n} K={k}: " + "  ".join(f"g{g}: {sorted(t)[len(t)//2]:7.1f}us ({sorted(t)[len(t)//2]/base:5.3f})" for g, t in times.items()), flush=True)
