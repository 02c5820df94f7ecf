#!/usr/bin/env python3
"""Dev tool: A/B of the persistent GEMM instance (WM_GEMM_PERSIST=0/1) in ONE process, interleaved rounds, and a bitwise
comparison of the two instances' outputs."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
M = a.batch * 4096
shapes = {"qkv": (M, 3840, 1280, 0), "lin1": (M, 5120, 1280, 1), "vitl_qkv": (M, 3072, 1024, 0)}
dev = G.dev()
for name, (m, n, k, act) in shapes.items():
    A = G.to16(torch.randn(m, k, device=dev), "bf16")
    W = G.to16(torch.randn(n, k, device=dev) / math.sqrt(k), "bf16")
    bias = torch.randn(n, device=dev)
    outs, times = {}, {"0": [], "1": []}
    def run():
        return G.gemm16(A, W, bias, act=act, prec="bf16", want32=False, want16=True)[1]
    for rnd in range(a.rounds):
        for mode in ("0", "1"):
            os.environ["WM_GEMM_PERSIST"] = mode
            outs[mode] = run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): run()
            e1.record(); torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) * 1e3 / a.iters)
    same = torch.equal(outs["0"], outs["1"])
    ref = A.float() @ W.float().t() + bias
    if act == 1:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
    err = ((outs["1"].float() - ref).norm() / ref.norm()).item()
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print(f"{name:9s} M={m} N={n} K={k}: plain {med['0']:7.1f} us  persistent {med['1']:7.1f} us ({med['1'] / med['0']:.3f})  bitwise equal: {same}  rel err vs fp32 {err:.2e}", flush=True)
