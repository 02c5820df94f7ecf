#!/usr/bin/env python3
"""Dev tool: time the decoder-side fp32 ops (attention shapes and small GEMMs) and check them against torch."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

dev = G.dev()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def ref_mha(q, k, v, heads):
    B, nq, C = q.shape; hd = C // heads
    qh = q.view(B, nq, heads, hd).transpose(1, 2).double(); kh = k.view(B, -1, heads, hd).transpose(1, 2).double(); vh = v.view(B, -1, heads, hd).transpose(1, 2).double()
    a = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(hd), -1)
    return (a @ vh).transpose(1, 2).reshape(B, nq, C).float()

B = int(os.environ.get("DEC_B", "4"))
for name, nq, nk, C in (("t2i", 51, 4096, 128), ("i2t", 4096, 51, 128), ("self", 51, 51, 256)):
    q = torch.randn(B, nq, C, device=dev); k = torch.randn(B, nk, C, device=dev); v = torch.randn(B, nk, C, device=dev)
    out = G.mha32(q, k, v, 8)
    err = (out - ref_mha(q, k, v, 8)).abs().max().item()
    print(f"mha32 {name:5s} nq={nq} nk={nk} C={C}: {timeit(lambda: G.mha32(q, k, v, 8)):8.1f} us  max err {err:.2e}", flush=True)
for name, M, N, K in (("tok 256->256", 51 * B, 256, 256), ("tok 256->128", 51 * B, 128, 256), ("tok 128->256", 51 * B, 256, 128), ("mlp1", 51 * B, 2048, 256),
                      ("mlp2", 51 * B, 256, 2048), ("head8", 51 * B, 8, 256), ("keys 256->128", 4096 * B, 128, 256), ("keys 128->256", 4096 * B, 256, 128)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / math.sqrt(K); bias = torch.randn(N, device=dev)
    out = G.gemm32(a, w, bias)
    err = (out - (a.double() @ w.double().t() + bias.double()).float()).abs().max().item()
    print(f"gemm32 {name:14s} M={M} N={N} K={K}: {timeit(lambda: G.gemm32(a, w, bias)):8.1f} us  max err {err:.2e}", flush=True)
