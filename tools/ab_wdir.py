#!/usr/bin/env python3
"""Dev tool (round 3): A/B of the GEMM with W fragments loaded straight into registers (WM_GEMM_WDIR=1)
against the default kernel (W through the LDS ring) in ONE process, interleaved rounds, random data, plus a bitwise
comparison of the two instances' outputs.  Shapes = the four block GEMMs as the model launches them (proj / lin2 with the
fp32 residual, lin1 with GELU)."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_util as G

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--prec", default="fp16")
ap.add_argument("--env", default="WM_GEMM_WDIR")
ap.add_argument("--value", default="1", help="1 = row-major W (16 rows x 64 B per load); 2 = W pre-packed in fragment order (1 KiB contiguous per load)")
a = ap.parse_args()
M = a.batch * 4096
# name: (M, N, K, act, residual)
shapes = {"qkv": (M, 3840, 1280, 0, False), "proj": (M, 1280, 1280, 0, True), "lin1": (M, 5120, 1280, 1, False), "lin2": (M, 1280, 5120, 0, True),
          "vitl_qkv": (M, 3072, 1024, 0, False)}
dev = G.dev()
tot = {"0": 0.0, "1": 0.0}
for name, (m, n, k, act, res) in shapes.items():
    A = G.to16(torch.randn(m, k, device=dev), a.prec)
    W = G.to16(torch.randn(n, k, device=dev) / math.sqrt(k), a.prec)
    bias = torch.randn(n, device=dev)
    R = torch.randn(m, n, device=dev) if res else None
    outs, times = {}, {"0": [], "1": []}
    Wp = W.view(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(n, k) if a.value == "2" else W
    if a.env == "WM_GEMM_WPACK":
        # LDS-image order of a DMA piece: position l (16-byte unit) holds row l >> 2, chunk (l & 3) ^ ((-(l >> 4)) & 3)
        l = torch.arange(64, device=dev)
        row, ch = l >> 2, (l & 3) ^ ((-(l >> 4)) & 3)
        Wv = W.view(n // 16, 16, k // 32, 4, 8)
        Wp = Wv[:, row, :, ch, :].permute(1, 2, 0, 3).contiguous().view(n, k)       # advanced indexing puts the 64 positions first
        Ap = A.view(m // 16, 16, k // 32, 4, 8)[:, row, :, ch, :].permute(1, 2, 0, 3).contiguous().view(m, k) if a.value == "2" else A
    if a.env != "WM_GEMM_WPACK":
        Ap = A
    cur = {"w": W, "a": A}
    def run():
        o32, o16 = G.gemm16(cur["a"], cur["w"], bias, residual=R, act=act, prec=a.prec, want32=res, want16=not res)
        return o32 if res else o16
    for rnd in range(a.rounds):
        for mode in ("0", "1"):
            os.environ[a.env] = "0" if mode == "0" else a.value
            cur["w"] = W if mode == "0" else Wp
            cur["a"] = A if mode == "0" else Ap
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
    if res:
        ref = ref + R
    err = ((outs["1"].float() - ref).norm() / ref.norm()).item()
    med = {k_: sorted(v)[len(v) // 2] for k_, v in times.items()}
    mn = {k_: min(v) for k_, v in times.items()}
    if name != "vitl_qkv":
        tot["0"] += med["0"]; tot["1"] += med["1"]
    tf = 2.0 * m * n * k / 1e6
    print(f"{name:9s} M={m} N={n} K={k}: default {med['0']:7.1f} us ({tf / med['0']:6.0f} TF, min {mn['0']:.1f})  {a.env}={a.value} {med['1']:7.1f} us ({tf / med['1']:6.0f} TF, min {mn['1']:.1f})"
          f"  ratio {med['1'] / med['0']:.3f}  bitwise equal: {same}  rel err vs fp32 {err:.2e}", flush=True)
print(f"sum of the four ViT-H block GEMMs: default {tot['0']:.1f} us, {a.env}=1 {tot['1']:.1f} us, ratio {tot['1'] / tot['0']:.4f}")
