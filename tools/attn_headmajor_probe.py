import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gpu_util as G
from wildlifemapper_amd import _native as N
dev = G.dev()
"""Dev probe: would a head-major q / k / v layout ([image][head][token][hd], each K / V tile contiguous) speed up the global attention
kernel?  The mha16 entry point takes per-tensor strides, so a head-major copy can be fed as (image x head) "images" of one head."""
def mha16s(q, k, v, stride, batch, heads, hd, out):
    N.check(N.lib().wm_op_mha16(N.ptr(q), stride, N.ptr(k), stride, N.ptr(v), stride, N.ptr(out), heads * hd, batch, heads, hd, 4096, 4096,
                                G.PRECS["fp16"][0], G.sp()))
    return out
B, heads, hd = 16, 16, 80
D = heads * hd
qkv = G.to16(torch.randn(B * 4096, 3 * D, device=dev) * 0.5, "fp16")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
# row-major (token-major) qkv, as the engine has it
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
out_rm = torch.empty(B * 4096, D, device=dev, dtype=torch.float16)
t_rm = timeit(lambda: mha16s(q, k, v, 3 * D, B, heads, hd, out_rm))
# head-major copies: [B*heads][4096][hd], each (image, head) a "batch" of one head
def hm(x): return x.view(B, 4096, heads, hd).permute(0, 2, 1, 3).contiguous().view(B * heads * 4096, hd)
qh, kh, vh = hm(q), hm(k), hm(v)
out_hm = torch.empty(B * heads * 4096, hd, device=dev, dtype=torch.float16)
t_hm = timeit(lambda: mha16s(qh, kh, vh, hd, B * heads, 1, hd, out_hm))
same = torch.equal(out_hm.view(B, heads, 4096, hd).permute(0, 2, 1, 3).reshape(B * 4096, D), out_rm)
print(f"global attention (no rel-pos) token-major qkv {t_rm:.1f} us, head-major q/k/v {t_hm:.1f} us, identical {same}")

# ---- window attention: token-major packed qkv vs head-major copies ----
bias = torch.randn(3 * D, device=dev) * 0.1
rel_h = torch.randn(27, hd, device=dev) * 0.1
rel_w = torch.randn(27, hd, device=dev) * 0.1
out_w = torch.empty(B * 4096, D, device=dev, dtype=torch.float16)
def win_tm():
    N.check(N.lib().wm_op_encoder_attention(N.ptr(qkv), N.ptr(bias), N.ptr(rel_h), N.ptr(rel_w), N.ptr(out_w), B, heads, hd, 14,
                                            G.PRECS["fp16"][0], G.sp()))
out_wh = torch.empty(B * heads * 4096, hd, device=dev, dtype=torch.float16)
bias1 = torch.cat([bias[0:hd], bias[D:D + hd], bias[2 * D:2 * D + hd]]).contiguous()      # head 0's rows as a one-head bias
def win_hm():
    N.check(N.lib().wm_op_encoder_attention_qkv(N.ptr(qh), N.ptr(kh), N.ptr(vh), hd, N.ptr(bias1), N.ptr(rel_h), N.ptr(rel_w), N.ptr(out_wh),
                                                B * heads, 1, hd, 14, G.PRECS["fp16"][0], G.sp()))
t_wt, t_wh = timeit(win_tm), timeit(win_hm)
a = out_w.view(B, 4096, heads, hd)[:, :, 0]; b = out_wh.view(B, heads, 4096, hd)[:, 0]
print(f"window attention token-major qkv {t_wt:.1f} us, head-major q/k/v {t_wh:.1f} us, head 0 identical {torch.equal(a, b)}")
