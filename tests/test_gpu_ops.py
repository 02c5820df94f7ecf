"""Kernel-level parity (-m gpu): every HIP kernel, through the C-ABI, against a plain
fp32 evaluation of the same operation on the same (already 16-bit-rounded) operands.

Tolerances (relative L2 unless stated):
  fp32-output kernels   1e-5   (fp32 accumulation order only)
  16-bit-output kernels bf16 5e-3 / fp16 7e-4 (one output rounding: 2^-9 / 2^-12)
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import wm_oracle as O          # checker only
import gpu_util as G

OUT16_TOL = {"bf16": 5e-3, "fp16": 7e-4}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device (run with -m 'not gpu' on CPU)")
    torch.manual_seed(0)


def _rnd(t, prec):
    return t.to(G.PRECS[prec][1]).float()


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_cvt_matches_torch_rounding(prec):
    x = torch.randn(1 << 16, device=G.dev()) * 3
    x[:4] = torch.tensor([0.0, -0.0, 1e-8, 70000.0], device=G.dev())
    y = G.to16(x, prec)
    want = x.clamp(-65504, 65504).to(torch.float16) if prec == "fp16" else x.to(torch.bfloat16)
    assert torch.equal(y.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("M,N,K,act", [(4096, 3840, 1280, 0), (4096, 5120, 1280, 1), (128, 128, 64, 0)])
def test_fp16_outputs_saturate_instead_of_overflowing(M, N, K, act):
    """fp16 is the default operand type; its range ends at 65504.  A 16-bit GEMM output that exceeds it (here |y| up to ~2e5,
    through the plain and the GELU epilogue of the block kernels and the small-shape kernel) must come out as +-65504, never
    inf / NaN, and in-range values must be unaffected."""
    dev = G.dev()
    a = G.to16(torch.randn(M, K, device=dev) * 200.0, "fp16")
    w = G.to16(torch.randn(N, K, device=dev) * 200.0 / math.sqrt(K), "fp16")
    o32, o16 = G.gemm16(a, w, None, act=act, prec="fp16", want32=True, want16=True)
    ref = a.float() @ w.float().t()
    if act == 1:
        ref = 0.5 * ref * (1 + torch.erf(ref / math.sqrt(2)))
    assert bool(torch.isfinite(o16.float()).all())
    assert (ref.abs() > 65504).sum().item() > 0                       # the case does overflow fp16
    want = ref.clamp(-65504, 65504)
    big = ref.abs() > 65504
    assert bool((o16.float()[big].abs() == 65504).all())
    assert G.rel_l2(o16.float()[~big], want[~big]) < 2e-3
    assert G.rel_l2(o32, ref) < 2e-3                                    # the fp32 output of the same launch is not clamped


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1280), (4096, 3840, 1280), (512, 256, 2304), (8192, 1280, 5120)])
def test_gemm16_plain(prec, M, N, K):
    a = G.to16(torch.randn(M, K, device=G.dev()), prec)
    w = G.to16(torch.randn(N, K, device=G.dev()) / math.sqrt(K), prec)
    o32, o16 = G.gemm16(a, w, prec=prec, want16=True)
    ref = a.float() @ w.float().t()
    assert G.rel_l2(o32, ref) < 1e-5
    assert G.rel_l2(o16.float(), ref) < OUT16_TOL[prec]


def test_gemm16_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed or permuted C write (cdna guide §3)."""
    n = 128
    a = torch.zeros(n, 128, device=G.dev())
    a[:, :n] = torch.eye(n, device=G.dev())
    w = (torch.arange(256 * 128, device=G.dev(), dtype=torch.float32).reshape(256, 128) % 251) - 125
    o32, _ = G.gemm16(G.to16(a, "bf16"), G.to16(w, "bf16"))
    assert torch.equal(o32, w[:, :n].t().contiguous())


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(384, 256, 320), (512, 256, 320), (8192, 1280, 192)])   # v1 tile, v2 BN=128, v2 BN=160
def test_gemm16_epilogue(prec, act, M, N, K):
    a = G.to16(torch.randn(M, K, device=G.dev()), prec)
    w = G.to16(torch.randn(N, K, device=G.dev()) / math.sqrt(K), prec)
    bias = torch.randn(N, device=G.dev())
    res = torch.randn(128, N, device=G.dev())          # broadcast residual (rows modulo 128)
    o32, o16 = G.gemm16(a, w, bias, res, 128, act, prec, want16=True)
    y = a.float() @ w.float().t() + bias
    if act == 1:
        y = O.gelu_erf(y)
    elif act == 2:
        y = torch.relu(y)
    y = y + res.repeat(M // 128, 1)
    assert G.rel_l2(o32, y) < 2e-5
    assert G.rel_l2(o16.float(), y) < OUT16_TOL[prec]


def test_gemm16_inplace_residual():
    M, N, K = 256, 128, 128
    a = G.to16(torch.randn(M, K, device=G.dev()), "bf16")
    w = G.to16(torch.randn(N, K, device=G.dev()) / 11, "bf16")
    x = torch.randn(M, N, device=G.dev())
    want = x + a.float() @ w.float().t()
    from wildlifemapper_amd import _native as Nn
    Nn.check(Nn.lib().wm_op_gemm16(Nn.ptr(a), Nn.ptr(w), None, Nn.ptr(x), 0, Nn.ptr(x), None, M, N, K, 0, 0, G.sp()))
    assert G.rel_l2(x, want) < 1e-5


def _variants_run(fn):
    """Run fn() and return {GEMM kernel instance: launches} for the instances it launched (wm_debug_gemm_variant_counts)."""
    from wildlifemapper_amd import _native as Nn
    Nn.gemm_variant_counts(reset=True)
    out = fn()
    torch.cuda.synchronize()
    return out, {k: v for k, v in Nn.gemm_variant_counts().items() if v}


def _residual_case(M, N, K, outs, prec):
    code, dt = G.PRECS[prec]
    a = G.to16(torch.randn(M, K, device=G.dev()), prec)
    w = G.to16(torch.randn(N, K, device=G.dev()) / math.sqrt(K), prec)
    bias = torch.randn(N, device=G.dev())
    # rows with distinct offsets / scales: a residual row landing in the wrong output row (or pass) would show
    x = torch.randn(M, N, device=G.dev()) * (0.5 + torch.rand(M, 1, device=G.dev())) + torch.arange(M, device=G.dev()).view(M, 1) % 97 * 0.25
    want = x + a.float() @ w.float().t() + bias
    from wildlifemapper_amd import _native as Nn
    o16 = torch.empty(M, N, device=G.dev(), dtype=dt) if outs != "f32" else None
    o32 = x if outs != "16" else None                        # in place, as the encoder uses it (image_encoder.py:200-203)
    _, var = _variants_run(lambda: Nn.check(Nn.lib().wm_op_gemm16(Nn.ptr(a), Nn.ptr(w), Nn.ptr(bias), Nn.ptr(x), 0, Nn.ptr(o32), Nn.ptr(o16),
                                                                  M, N, K, 0, code, G.sp())))
    if o32 is not None:
        assert G.rel_l2(o32, want) < 1e-5
        assert (o32 - want).abs().max().item() < 1e-3      # no single misplaced element hides inside an L2 norm
    if o16 is not None:
        assert G.rel_l2(o16.float(), want) < OUT16_TOL[prec]
    return var


@pytest.mark.parametrize("N,variant", [(640, "v2_160"), (512, "v2_128")])
@pytest.mark.parametrize("outs", ["f32", "both", "16"])
def test_gemm16_residual_paths_few_tiles(N, variant, outs):
    """Residual epilogue of the half-width kernel (gemm16_v2.h), which the dispatch picks when the 256 x 320 / 256 tiles
    would leave most CUs idle (one or two image tiles per call)."""
    assert _residual_case(768, N, 256, outs, "bf16") == {variant: 1}


@pytest.mark.parametrize("M,N,K,variant", [(16384, 1280, 1280, "v5_320_res"),      # ViT-H proj at B = 4: 256 workgroups, one round
                                           (16384, 1280, 5120, "v5_320_res"),      # ViT-H lin2 at B = 4
                                           (65536, 1280, 1280, "v5_320_res"),      # proj at B = 16 (configs[2]): 4 rounds
                                           (16384, 1024, 1024, "v5_256_res"),      # ViT-L proj / HFC adaptor out_proj
                                           (12288, 768, 3072, "v5_256_res")])      # ViT-B lin2 at B = 3
@pytest.mark.parametrize("outs", ["f32", "both", "16"])
def test_gemm16_v5_residual_reaches_kernel(M, N, K, variant, outs):
    """The staggered kernel's residual epilogue (gemm16_v5.h: residual tile by LDS-DMA, 8 passes, one pass ahead) at
    the shapes the timed ViT-H configurations launch, against fp32; the instance that ran is asserted, so a change of the
    dispatch heuristic cannot silently un-test it (round-1 VERDICT weak #1)."""
    assert _residual_case(M, N, K, outs, "bf16") == {variant: 1}


def test_gemm16_v5_residual_fp16_and_broadcast():
    assert _residual_case(16384, 1280, 1280, "both", "fp16") == {"v5_320_res": 1}
    # broadcast residual (pos_embed add of the patch embed, image_encoder.py:125-126): rows modulo 4096
    M, N, K = 16384, 1280, 768
    a = G.to16(torch.randn(M, K, device=G.dev()), "fp16")
    w = G.to16(torch.randn(N, K, device=G.dev()) / math.sqrt(K), "fp16")
    bias = torch.randn(N, device=G.dev())
    res = torch.randn(4096, N, device=G.dev())
    (o32, o16), var = _variants_run(lambda: G.gemm16(a, w, bias, res, 4096, 0, "fp16", want16=True))
    assert var == {"v5_320_res": 1}
    want = a.float() @ w.float().t() + bias + res.repeat(4, 1)
    assert G.rel_l2(o32, want) < 1e-5 and G.rel_l2(o16.float(), want) < OUT16_TOL["fp16"]


@pytest.mark.parametrize("M,N,K,act,variant", [(16384, 3840, 1280, 0, "v5_320"),     # ViT-H qkv, B = 4
                                               (16384, 5120, 1280, 1, "v5_320"),     # lin1 + GELU
                                               (4096, 5120, 1280, 1, "v5_320"),      # lin1 at B = 1: still the wide tile
                                               (4096, 1280, 5120, 0, "v2_160"),      # lin2 at B = 1 -> half-width kernel
                                               (16384, 1024, 1024, 2, "v5_256"),     # HFC adaptor linear1 + ReLU
                                               (16384, 256, 1280, 0, "v2_128")])     # neck 1x1: 64 tiles
def test_gemm16_dispatch_and_values_at_model_shapes(M, N, K, act, variant):
    prec = "bf16"
    a = G.to16(torch.randn(M, K, device=G.dev()), prec)
    w = G.to16(torch.randn(N, K, device=G.dev()) / math.sqrt(K), prec)
    bias = torch.randn(N, device=G.dev())
    (o32, o16), var = _variants_run(lambda: G.gemm16(a, w, bias, None, 0, act, prec, want16=True))
    assert var == {variant: 1}
    y = a.float() @ w.float().t() + bias
    y = {0: y, 1: O.gelu_erf(y), 2: torch.relu(y)}[act]
    assert G.rel_l2(o32, y) < 2e-5
    assert G.rel_l2(o16.float(), y) < OUT16_TOL[prec]


def test_pack16_is_the_documented_permutation():
    """wm_op_pack16 (the kernel that packs the weights at wm_finalize_weights) against the layout's definition restated with
    torch indexing, and the inverse."""
    from wildlifemapper_amd import _native as Nn
    t = torch.arange(48 * 96, device=G.dev(), dtype=torch.int32).to(torch.int16).view(48, 96)
    p = G.pack16(t)
    assert torch.equal(p, G.pack16_torch(t))
    assert torch.equal(G.unpack16_torch(p), t)
    w = G.to16(torch.randn(1280, 5120, device=G.dev()), "fp16")
    assert torch.equal(G.unpack16_torch(G.pack16(w)), w)
    with pytest.raises(RuntimeError, match="pack16"):
        G.pack16(torch.zeros(24, 64, device=G.dev(), dtype=torch.float16))


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,act,res", [(16384, 3840, 1280, 0, False),      # qkv: norm1 output packed, weight packed
                                           (16384, 5120, 1280, 1, False),      # lin1 + GELU, output packed for lin2
                                           (16384, 1280, 5120, 0, True),       # lin2: packed GELU hidden in, fp32 residual out
                                           (16384, 1024, 1024, 2, False),      # 256-wide tile instance (ViT-L / HFC adaptor shapes)
                                           (65536, 1280, 1280, 0, True)])      # proj at B = 16: packed weight only
def test_gemm16_packed_operands_bit_identical(M, N, K, act, res, prec):
    """Operands in LDS-image order (round 3: every DMA piece 8 whole lines instead of 16 half lines) put the same bytes in the
    same LDS places, so every combination of packed W / packed A / packed 16-bit output must equal the row-major launch bit
    for bit (and the packed output, un-permuted, the row-major output)."""
    from wildlifemapper_amd import _native as Nn
    dev = G.dev()
    assert Nn.lib().wm_op_gemm16_takes_packed(M, N, K) == 1
    a = G.to16(torch.randn(M, K, device=dev), prec)
    w = G.to16(torch.randn(N, K, device=dev) / math.sqrt(K), prec)
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev) if res else None
    ap, wp = G.pack16(a), G.pack16(w)
    base32, base16 = G.gemm16(a, w, bias, r, 0, act, prec, want32=res, want16=True)
    y = a.float() @ w.float().t() + bias
    y = {0: y, 1: O.gelu_erf(y), 2: torch.relu(y)}[act] + (r if res else 0)
    assert G.rel_l2(base16.float(), y) < OUT16_TOL[prec]
    for layout, aa, ww in ((Nn.GEMM_W_PACKED, a, wp), (Nn.GEMM_A_PACKED, ap, w), (Nn.GEMM_W_PACKED | Nn.GEMM_A_PACKED, ap, wp)):
        o32, o16 = G.gemm16(aa, ww, bias, r, 0, act, prec, want32=res, want16=True, layout=layout)
        assert torch.equal(o16, base16), layout
        if res:
            assert torch.equal(o32, base32), layout
    if not res:
        _, o16p = G.gemm16(ap, wp, bias, None, 0, act, prec, want32=False, want16=True,
                           layout=Nn.GEMM_W_PACKED | Nn.GEMM_A_PACKED | Nn.GEMM_OUT_PACKED)
        assert torch.equal(G.unpack16_torch(o16p), base16)
    else:
        with pytest.raises(RuntimeError, match="packed output"):
            G.gemm16(a, w, bias, r, 0, act, prec, want32=True, want16=True, layout=Nn.GEMM_OUT_PACKED)


def test_packed_layout_rejected_where_the_half_width_kernel_runs():
    from wildlifemapper_amd import _native as Nn
    dev = G.dev()
    M, N, K = 4096, 1280, 5120                           # lin2 at B = 1: 64 wide tiles -> gemm16_v2<160>
    assert Nn.lib().wm_op_gemm16_takes_packed(M, N, K) == 0
    a = G.to16(torch.randn(M, K, device=dev), "fp16")
    w = G.to16(torch.randn(N, K, device=dev), "fp16")
    with pytest.raises(RuntimeError, match="row-major operands only"):
        G.gemm16(a, w, None, None, 0, 0, "fp16", layout=Nn.GEMM_W_PACKED)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("C", [1280, 1024, 768])
def test_layernorm_packed_output(prec, C):
    """The blocks' LayerNorm writing its 16-bit output in LDS-image order (the A operand of qkv / lin1): the same values,
    permuted."""
    dev = G.dev()
    x = torch.randn(8192, C, device=dev) * 2 + 0.3
    g, b = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
    _, plain = G.layernorm(x, g, b, 1e-6, prec, want32=False, want16=True)
    _, packed = G.layernorm(x, g, b, 1e-6, prec, want32=False, want16=True, packed=True)
    assert torch.equal(G.unpack16_torch(packed), plain)
    assert torch.equal(packed, G.pack16(plain))


# ---------------------------------------------------------------------------
# Folded LayerNorm (WM_CFG_FOLD_LN): statistics in the producing GEMM, normalisation in the consuming GEMM
# ---------------------------------------------------------------------------
def _outlier_rows(M, C, dev):
    """Residual-stream-like rows: unit bulk, a per-row offset, two massive channels (~200x) as the outlier weight profile makes."""
    x = torch.randn(M, C, device=dev) * (0.7 + torch.rand(M, 1, device=dev)) + 0.2 * torch.randn(M, 1, device=dev)
    x[:, 101] += 190.0
    x[:, 678 % C] -= 240.0
    return x


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("C", [1280, 1024, 768])
def test_ln_stats16_partials_and_copy(prec, C):
    dev = G.dev()
    x = _outlier_rows(4096, C, dev)
    stats, x16 = G.ln_stats16(x, prec)
    bn = G.fold_bn(C)
    xt = x.double().view(4096, C // bn, bn)
    mean = xt.mean(-1)
    m2 = ((xt - mean[..., None]) ** 2).sum(-1)
    assert (stats[..., 0].double() - mean).abs().max().item() < 1e-5 * max(1.0, mean.abs().max().item())
    assert ((stats[..., 1].double() - m2).abs() / m2).max().item() < 1e-5
    assert torch.equal(G.unpack16_torch(x16), G.to16(x, prec))


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_fold_weight16(prec):
    dev = G.dev()
    Nn, K = 768, 1280
    w = G.to16(torch.randn(Nn, K, device=dev) / math.sqrt(K), prec)
    g, b, bias = 1 + 0.3 * torch.randn(K, device=dev), 0.2 * torch.randn(K, device=dev), torch.randn(Nn, device=dev)
    g[5] = 50.0
    wf, c1, c2 = G.fold_weight16(w, g, b, bias, prec)
    want_wf = G.to16(w.float() * g, prec)
    assert torch.equal(G.unpack16_torch(wf), want_wf)
    assert (c1.double() - want_wf.double().sum(1)).abs().max().item() < 1e-4
    assert (c2.double() - ((w.double() * b.double()).sum(1) + bias.double())).abs().max().item() < 1e-4


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K,act", [(16384, 3840, 1280, 0), (16384, 5120, 1280, 1), (12288, 3072, 768, 1), (16384, 3072, 1024, 0)])
def test_gemm16_folded_equals_layernorm_then_gemm(M, N, K, act, prec):
    """LN(x) W^T + b computed as rstd (x16 (gamma W)^T - mean c1) + c2 in the GEMM epilogue, against fp32 torch, and no worse
    than the classic route (LayerNorm kernel -> 16-bit -> GEMM) on rows with massive channels and a per-row offset."""
    dev = G.dev()
    x = _outlier_rows(M, K, dev)
    w = G.to16(torch.randn(N, K, device=dev) / math.sqrt(K), prec)
    g, b, bias = 1 + 0.2 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev), torch.randn(N, device=dev)
    g[17], g[300] = 50.0, -30.0
    want = torch.nn.functional.layer_norm(x, (K,), g, b, 1e-6) @ w.float().t() + bias
    want = O.gelu_erf(want) if act == 1 else want
    stats, x16 = G.ln_stats16(x, prec)
    wf, c1, c2 = G.fold_weight16(w, g, b, bias, prec)
    got = G.gemm16_folded(x16, wf, c1, c2, stats, 1e-6, act, prec)
    e_fold = G.rel_l2(got.float(), want)
    _, xn16 = G.layernorm(x, g, b, 1e-6, prec, want32=False, want16=True)
    _, classic = G.gemm16(xn16, w, bias, None, 0, act, prec, want32=False, want16=True)
    e_classic = G.rel_l2(classic.float(), want)
    print(f"[fold {prec} M={M} N={N} K={K} act={act}] folded {e_fold:.2e}  classic {e_classic:.2e}")
    assert e_fold < 2.5 * OUT16_TOL[prec] and e_fold < 2.0 * e_classic + 1e-4
    got_p = G.gemm16_folded(x16, wf, c1, c2, stats, 1e-6, act, prec, out_packed=True)
    assert torch.equal(G.unpack16_torch(got_p), got)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(16384, 1280, 1280), (16384, 1280, 5120), (65536, 1280, 1280), (16384, 1024, 1024), (12288, 768, 3072)])
def test_gemm16_stats_producer_is_the_residual_gemm_plus_the_standalone_statistics(M, N, K, prec):
    """The producing GEMM (FOLDP instance) writes the same fp32 rows as the plain residual GEMM, and the statistics / 16-bit
    copy it adds are bit-identical to what the standalone kernel computes from those rows: a tile's folded LayerNorm does not
    depend on which of the two produced them (i.e. on the batch size)."""
    dev = G.dev()
    a = G.to16(torch.randn(M, K, device=dev), prec)
    w = G.to16(torch.randn(N, K, device=dev) / math.sqrt(K), prec)
    bias = torch.randn(N, device=dev)
    r = _outlier_rows(M, N, dev)
    base32, _ = G.gemm16(a, w, bias, r, 0, 0, prec, want32=True, want16=False)
    (out32, x16, stats), var = _variants_run(lambda: G.gemm16_stats(a, w, bias, r, prec))
    assert var == {("v5_320_foldp" if N % 320 == 0 else "v5_256_foldp"): 1}
    assert torch.equal(out32, base32)
    st2, x2 = G.ln_stats16(out32, prec)
    assert torch.equal(stats, st2) and torch.equal(x16, x2)
    from wildlifemapper_amd import _native as Nn
    o2, x3, st3 = G.gemm16_stats(G.pack16(a), G.pack16(w), bias, r, prec, layout=Nn.GEMM_W_PACKED | Nn.GEMM_A_PACKED)
    assert torch.equal(o2, base32) and torch.equal(x3, x16) and torch.equal(st3, stats)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("C", [1280, 1024, 768])
def test_split_stream_planes_and_merge(prec, C):
    """Split stream (gemm16_v5.h): hi = round16(x) in LDS-image order, lo = fp16(x - hi), the rewritten fp32 rows = float(hi) +
    float(lo) = what stream_merge reconstructs; statistics and hi equal the unsplit producer's; x is recovered to 2^-19 (bf16 hi) /
    2^-21 (fp16 hi) relative; unpack16 inverts pack16."""
    dev = G.dev()
    x = _outlier_rows(4096, C, dev)
    st0, x16 = G.ln_stats16(x, prec)
    stats, hi, lo, xr = G.ln_stats16_split(x, prec)
    assert torch.equal(stats, st0) and torch.equal(hi, x16)
    hi_rm, lo_rm = G.unpack16(hi), G.unpack16(lo)
    assert torch.equal(hi_rm, G.unpack16_torch(hi)) and torch.equal(hi_rm, G.to16(x, prec))
    assert torch.equal(lo_rm, (x - hi_rm.float()).to(torch.float16))
    assert torch.equal(xr, hi_rm.float() + lo_rm.float())
    assert torch.equal(G.stream_merge(hi, lo, prec), xr)
    # x = hi + lo to 2^-19 (bf16 hi: 8 + 11 bits) / 2^-21 (fp16 hi: 11 + 11 bits) relative, down to fp16's subnormal spacing 2^-24
    bound = x.abs() * (2.0 ** -18 if prec == "bf16" else 2.0 ** -20) + 2.0 ** -24
    assert bool(((xr - x).abs() <= bound).all()), ((xr - x).abs() - bound).max().item()


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(16384, 1280, 1280), (16384, 1280, 5120), (65536, 1280, 1280), (16384, 1024, 1024), (12288, 768, 3072)])
def test_gemm16_split_stream_producer_equals_fp32_path(M, N, K, prec):
    """The SPLIT instance (residual planes in and out by LDS-DMA) against the fp32-stream path on the same rounded stream:
    v = residual + a w^T + bias from the plain residual GEMM, then the standalone kernel's statistics / hi / lo of v -- bit for
    bit what the split GEMM writes in place.  This is the identity that makes a tile independent of its batch size (a small call
    keeps the stream in fp32 and rounds it in the standalone kernel)."""
    dev = G.dev()
    a = G.to16(torch.randn(M, K, device=dev), prec)
    w = G.to16(torch.randn(N, K, device=dev) / math.sqrt(K), prec)
    bias = torch.randn(N, device=dev)
    _, hi, lo, xr = G.ln_stats16_split(_outlier_rows(M, N, dev), prec)
    v32, _ = G.gemm16(a, w, bias, xr, 0, 0, prec, want32=True, want16=False)
    st_ref, hi_ref, lo_ref, _ = G.ln_stats16_split(v32, prec)
    (hi2, lo2, stats), var = _variants_run(lambda: G.gemm16_split(a, w, bias, hi, lo, prec))
    assert var == {("v5_320_split" if N % 320 == 0 else "v5_256_split"): 1}
    assert torch.equal(stats, st_ref) and torch.equal(hi2, hi_ref) and torch.equal(lo2, lo_ref)
    from wildlifemapper_amd import _native as Nn
    hi3, lo3, st3 = G.gemm16_split(G.pack16(a), G.pack16(w), bias, hi, lo, prec, layout=Nn.GEMM_W_PACKED | Nn.GEMM_A_PACKED)
    assert torch.equal(hi3, hi_ref) and torch.equal(lo3, lo_ref) and torch.equal(st3, st_ref)


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
@pytest.mark.parametrize("B,Cin,N", [(2, 3, 1280), (1, 1, 1024), (3, 3, 768)])
def test_patch_embed_implicit_gemm(B, Cin, N, prec):
    """PatchEmbed / HfcEmbed (image_encoder.py:386-450: Conv2d k16 s16 + NCHW -> NHWC) as an implicit GEMM that gathers the
    patches from the 16-bit NCHW image (gemm16_v3.h AMODE 2, no im2col buffer) against F.conv2d on the same rounded operands."""
    dev = G.dev()
    torch.manual_seed(B * 7 + Cin)
    x = torch.randn(B, Cin, 1024, 1024, device=dev)
    w = torch.randn(N, Cin, 16, 16, device=dev) / math.sqrt(Cin * 256)
    bias = torch.randn(N, device=dev)
    x16, w16 = G.to16(x, prec), G.to16(w.reshape(N, -1), prec)
    o32, o16 = G.patch_embed16(x16, w16, bias, prec)
    ref = torch.nn.functional.conv2d(x16.float(), w16.float().view(N, Cin, 16, 16), bias, stride=16).permute(0, 2, 3, 1).reshape(B * 4096, N)
    assert G.rel_l2(o32, ref) < 2e-6
    assert torch.equal(o16, G.to16(o32, prec))


def test_gemm16_kernels_agree_bitwise():
    """A tile's result must not depend on which GEMM kernel its batch size selects (batch-invariant
    bit for bit): the same rows through the half-width kernel (M = 4096) and through the staggered 256 x 320 kernel
    (M = 16384), with GELU and with the fp32 residual."""
    prec, dev = "bf16", G.dev()
    K, N = 5120, 1280
    a = G.to16(torch.randn(16384, K, device=dev), prec)
    w = G.to16(torch.randn(N, K, device=dev) / math.sqrt(K), prec)
    bias = torch.randn(N, device=dev)
    res = torch.randn(16384, N, device=dev)
    (big32, big16), v1 = _variants_run(lambda: G.gemm16(a, w, bias, res, 0, 0, prec, want16=True))
    (sm32, sm16), v2 = _variants_run(lambda: G.gemm16(a[:4096].contiguous(), w, bias, res[:4096].contiguous(), 0, 0, prec, want16=True))
    assert v1 == {"v5_320_res": 1} and v2 == {"v2_160": 1}
    assert torch.equal(big32[:4096], sm32) and torch.equal(big16[:4096], sm16)
    # GELU epilogue: lin1-shaped, half-width (M = 256: 16 tiles) against the wide tile
    K, N = 1280, 5120
    a = G.to16(torch.randn(4096, K, device=dev), prec)
    w = G.to16(torch.randn(N, K, device=dev) / math.sqrt(K), prec)
    bias = torch.randn(N, device=dev)
    (_, big16), v1 = _variants_run(lambda: G.gemm16(a, w, bias, None, 0, 1, prec, want32=False, want16=True))
    (_, sm16), v2 = _variants_run(lambda: G.gemm16(a[:256].contiguous(), w, bias, None, 0, 1, prec, want32=False, want16=True))
    assert v1 == {"v5_320": 1} and v2 == {"v2_160": 1}
    assert torch.equal(big16[:256], sm16)


def test_gelu_fast_accuracy():
    """The 16-bit GEMM epilogue's GELU (degree-3/3 rational erf, wm_common.h) against the exact-erf form, fp32 output:
    C[m][n] = a[m][0] * w[n][0] with power-of-two row scales sweeps x over [-9, 9] exactly."""
    M, N, K = 256, 640, 64
    a = torch.zeros(M, K, device=G.dev())
    w = torch.zeros(N, K, device=G.dev())
    a[:, 0] = torch.tensor([2.0 ** (-(i % 4)) for i in range(M)], device=G.dev())
    w[:, 0] = torch.linspace(-9.0, 9.0, N, device=G.dev())
    a16, w16 = G.to16(a, "fp16"), G.to16(w, "fp16")
    o32, _ = G.gemm16(a16, w16, act=1, prec="fp16")
    x = a16.float() @ w16.float().t()
    assert (o32 - O.gelu_erf(x)).abs().max().item() < 2e-5


def test_gemm16_rejects_ragged():
    a = G.to16(torch.randn(100, 64, device=G.dev()), "bf16")
    w = G.to16(torch.randn(128, 64, device=G.dev()), "bf16")
    with pytest.raises(RuntimeError, match="multiples"):
        G.gemm16(a, w)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_conv3x3_implicit_gemm(prec):
    """Neck 3x3 conv (image_encoder.py:113-119) without an im2col buffer, against F.conv2d on the rounded operands."""
    B, Cin, Cout, dev = 2, 256, 256, G.dev()
    x = G.to16(torch.randn(B, 64, 64, Cin, device=dev), prec)                       # NHWC
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(9 * Cin)
    w16 = G.to16(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous(), prec)   # [co][tap][ci]
    out = torch.empty(B * 4096, Cout, device=dev)
    from wildlifemapper_amd import _native as Nn
    Nn.check(Nn.lib().wm_op_conv3x3_16(Nn.ptr(x), Nn.ptr(w16), Nn.ptr(out), B, Cout, Cin, G.PRECS[prec][0], G.sp()))
    wr = w16.float().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1).reshape(B * 4096, Cout)
    assert G.rel_l2(out, ref) < 1e-5
    # borders: the corner pixel sees only 4 taps
    assert torch.allclose(out[0], ref[0], rtol=1e-4, atol=1e-5) and torch.allclose(out[4095], ref[4095], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,N,K,act", [(51, 8, 256, 0), (102, 4, 256, 3), (4096, 128, 256, 0), (51, 2048, 256, 2), (153, 256, 2048, 0), (64, 64, 16, 1)])
def test_gemm32(M, N, K, act):
    a = torch.randn(M, K, device=G.dev())
    w = torch.randn(N, K, device=G.dev()) / math.sqrt(K)
    bias = torch.randn(N, device=G.dev())
    res = torch.randn(M, N, device=G.dev())
    out = G.gemm32(a, w, bias, res, act)
    y = (a.double() @ w.double().t() + bias.double())
    y = {0: y, 1: 0.5 * y * (1 + torch.erf(y / math.sqrt(2))), 2: torch.relu(y), 3: torch.sigmoid(y)}[act] + res.double()
    assert G.rel_l2(out, y) < 2e-6


@pytest.mark.parametrize("M,N,K,act", [(51, 8, 256, 0), (102, 4, 256, 3), (4096, 128, 256, 0), (51, 2048, 256, 2), (153, 256, 2048, 0),
                                       (816, 256, 256, 0), (65536, 128, 256, 0), (65536, 256, 128, 0), (64, 64, 32, 1), (70, 14, 96, 0)])
def test_gemm32_split_form(M, N, K, act):
    """The decoder's GEMMs since round 4: fp32 in and out on the 16-bit matrix pipe, every operand value split into fp16 hi + lo
    (gemm32.h gemm32x3_kernel).  Same bound against float64 as the fp32-MFMA kernel; decoder-sized weights (|w| ~ 0.02) and
    activations spanning 1e-3 .. 30; ragged M and N; one value per operand that needs the lo part to survive at all."""
    dev = G.dev()
    a = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev) * 1.5)         # rows of very different size
    w = torch.randn(N, K, device=dev) * 0.02
    a[0, 0], w[0, 0] = 1.0 + 2.0 ** -12, 1.0 - 2.0 ** -13                                    # not representable in fp16
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    out = G.gemm32(a, w, bias, res, act, split=True)
    y = (a.double() @ w.double().t() + bias.double())
    y = {0: y, 1: 0.5 * y * (1 + torch.erf(y / math.sqrt(2))), 2: torch.relu(y), 3: torch.sigmoid(y)}[act] + res.double()
    assert G.rel_l2(out, y) < 2e-6
    ref32 = G.gemm32(a, w, bias, res, act)
    print(f"gemm32 split form M={M} N={N} K={K}: rel-L2 vs float64 {G.rel_l2(out, y):.2e} (fp32 MFMA kernel: {G.rel_l2(ref32, y):.2e})")
    # per-row error relative to the row's own scale (a small row must not inherit a large row's error)
    rowerr = ((out.double() - y).norm(dim=1) / (y - res.double()).norm(dim=1).clamp_min(1e-30)).max().item()
    assert rowerr < 1e-5, rowerr


def test_gemm32_split_form_flags_fp16_range():
    """|a| >= 65504 cannot be split into fp16 parts: the handle-less op has no overflow word to raise, so the result is inf / nan,
    never a silently clamped number."""
    dev = G.dev()
    a = torch.ones(64, 32, device=dev)
    a[3, 5] = 1e5
    w = torch.ones(64, 32, device=dev) * 0.01
    out = G.gemm32(a, w, None, None, 0, split=True)
    assert not torch.isfinite(out[3]).all()
    assert torch.isfinite(out[4]).all()


@pytest.mark.parametrize("C,eps", [(256, 1e-5), (768, 1e-6), (1024, 1e-5), (1280, 1e-6)])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_layernorm(C, eps, prec):
    rows = 1000
    x = torch.randn(rows, C, device=G.dev()) * 3 + 0.7
    g = torch.rand(C, device=G.dev()) + 0.5
    b = torch.randn(C, device=G.dev())
    o32, o16 = G.layernorm(x, g, b, eps, prec, want16=True)
    ref = O.layer_norm(x.cpu(), g.cpu(), b.cpu(), eps)
    assert (o32.cpu() - ref).abs().max().item() < 2e-5
    assert G.rel_l2(o16.float(), ref) < OUT16_TOL[prec]


# ---------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------
def _ref_encoder_attention(qkv, bias16, rel_h, rel_w, B, heads, hd, window, prec):
    """fp32 evaluation of image_encoder.py:246-262 (+ window partition :190-199) from a 16-bit-rounded
    packed qkv; padded tokens carry the (16-bit rounded) qkv bias, P is rounded before P.V like the kernel."""
    D = heads * hd
    x = qkv.float().reshape(B, 64, 64, 3 * D)
    if window:
        xw, n = O.to_windows(x - bias16, window)
        xw = xw + bias16
        S = window
    else:
        xw, n, S = x, 0, 64
    Bp = xw.shape[0]
    t = xw.reshape(Bp, S * S, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    Rh = O.rel_pos_table(S, _rnd(rel_h, prec))
    Rw = O.rel_pos_table(S, _rnd(rel_w, prec))
    a = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    rq = q.reshape(Bp, heads, S, S, hd)
    rh = torch.einsum("bnhwc,hkc->bnhwk", rq, Rh)
    rw = torch.einsum("bnhwc,wkc->bnhwk", rq, Rw)
    a = (a.view(Bp, heads, S, S, S, S) + rh[..., :, None] + rw[..., None, :]).view(Bp, heads, S * S, S * S)
    p = _rnd(a.softmax(-1), prec)
    o = (p @ v).permute(0, 2, 1, 3).reshape(Bp, S, S, D)
    if window:
        o = O.from_windows(o, window, n, 64)
    return o.reshape(B * 4096, D)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("heads,hd", [(2, 80), (3, 64)])
def test_window_attention(prec, heads, hd):
    B, D = 2, heads * hd
    dev = G.dev()
    qkv = G.to16(torch.randn(B * 4096, 3 * D, device=dev), prec)
    bias = torch.randn(3 * D, device=dev) * 0.5
    rel_h = torch.randn(27, hd, device=dev) * 0.3
    rel_w = torch.randn(27, hd, device=dev) * 0.3
    out = G.encoder_attention(qkv, bias, rel_h, rel_w, B, heads, hd, 14, prec)
    ref = _ref_encoder_attention(qkv, _rnd(bias, prec), rel_h, rel_w, B, heads, hd, 14, prec)
    assert G.rel_l2(out.float(), ref) < 2 * OUT16_TOL[prec]
    assert (out.float() - ref).abs().max().item() < (0.06 if prec == "bf16" else 0.01)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("heads,hd", [(2, 80), (2, 64)])
def test_global_attention_relpos(prec, heads, hd):
    B, D = 1, heads * hd
    dev = G.dev()
    qkv = G.to16(torch.randn(B * 4096, 3 * D, device=dev), prec)
    bias = torch.zeros(3 * D, device=dev)
    rel_h = torch.randn(127, hd, device=dev) * 0.3
    rel_w = torch.randn(127, hd, device=dev) * 0.3
    out = G.encoder_attention(qkv, bias, rel_h, rel_w, B, heads, hd, 0, prec)
    ref = _ref_encoder_attention(qkv, _rnd(bias, prec), rel_h, rel_w, B, heads, hd, 0, prec)
    assert G.rel_l2(out.float(), ref) < 2 * OUT16_TOL[prec]


def test_global_attention_forces_rescale():
    """One key row spiked against the queries so the running max jumps at a late tile (online-softmax
    rescale branch; cdna guide rule 26)."""
    heads, hd, prec, dev = 1, 80, "bf16", G.dev()
    D = heads * hd
    x = torch.randn(4096, 3 * D, device=dev) * 0.3
    x[3000, D:2 * D] = x[100, 0:D] * 40          # key 3000 aligned with query 100
    qkv = G.to16(x, prec)
    zero = torch.zeros(3 * D, device=dev)
    rel = torch.zeros(127, hd, device=dev)
    out = G.encoder_attention(qkv, zero, rel, rel, 1, heads, hd, 0, prec)
    ref = _ref_encoder_attention(qkv, zero, rel, rel, 1, heads, hd, 0, prec)
    assert G.rel_l2(out.float(), ref) < 2 * OUT16_TOL[prec]
    assert (out.float()[100] - ref[100]).abs().max().item() < 0.05


@pytest.mark.parametrize("window", [14, 0])
def test_encoder_attention_qkv_three_tensors(window):
    """wm_op_encoder_attention_qkv: q / k / v as three tensors of one token stride.  Fed with per-head contiguous copies (each
    (image, head) as an image of one head) it must give the packed-qkv entry point's bits for that head."""
    B, heads, hd, prec = 2, 3, 80, "fp16"
    D, dev = heads * hd, G.dev()
    torch.manual_seed(5)
    qkv = G.to16(torch.randn(B * 4096, 3 * D, device=dev) * 0.5, prec)
    bias = torch.randn(3 * D, device=dev) * 0.1
    S = 14 if window else 64
    rel_h = torch.randn(2 * S - 1, hd, device=dev) * 0.2
    rel_w = torch.randn(2 * S - 1, hd, device=dev) * 0.2
    ref = G.encoder_attention(qkv, bias, rel_h, rel_w, B, heads, hd, window, prec).view(B, 4096, heads, hd)
    def per_head(x):
        return x.reshape(B, 4096, heads, hd).permute(0, 2, 1, 3).contiguous().view(B * heads * 4096, hd)
    q, k, v = per_head(qkv[:, :D]), per_head(qkv[:, D:2 * D]), per_head(qkv[:, 2 * D:])
    for head in range(heads):
        b1 = torch.cat([bias[head * hd:(head + 1) * hd], bias[D + head * hd:D + (head + 1) * hd],
                        bias[2 * D + head * hd:2 * D + (head + 1) * hd]]).contiguous()
        out = torch.empty(B * heads * 4096, hd, device=dev, dtype=torch.float16)
        N = G.N
        N.check(N.lib().wm_op_encoder_attention_qkv(N.ptr(q), N.ptr(k), N.ptr(v), hd, N.ptr(b1), N.ptr(rel_h), N.ptr(rel_w), N.ptr(out),
                                                    B * heads, 1, hd, window, G.PRECS[prec][0], G.sp()))
        assert torch.equal(out.view(B, heads, 4096, hd)[:, head], ref[:, :, head])


_ATTN_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import gpu_util as G
torch.manual_seed(11)
dev = G.dev()
for prec, hd, rel in (("fp16", 80, True), ("bf16", 64, True), ("fp16", 80, False)):
    heads, B = 2, 2
    D = heads * hd
    x = torch.randn(B * 4096, 3 * D, device=dev) * 0.5
    x[3000, D:2 * D] = x[100, 0:D] * 40                     # a late running-max jump: the deferred-rescale branch
    qkv = G.to16(x, prec)
    if rel:
        out = G.encoder_attention(qkv, torch.zeros(3 * D, device=dev), torch.randn(127, hd, device=dev) * 0.3,
                                  torch.randn(127, hd, device=dev) * 0.3, B, heads, hd, 0, prec)
    else:
        out = G.mha16(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, heads, hd, 4096, 4096, prec)
    torch.cuda.synchronize()
    import hashlib
    # without rel-pos the 8-wave kernel carries -m through the matrix pipe (bias k-step) and the 4-wave kernel adds it per score: the
    # same value to ~2^-22, not the same bits -- those cases print a checksum-free line and are compared numerically by the parent
    tag = hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest() if rel else "%.6f" % out.float().abs().mean().item()
    print(prec, hd, rel, tag, out.numel())
    if not rel:
        torch.save(out.cpu(), sys.argv[2] + "/norel_%s_%d.pt" % (prec, hd))
"""


def test_global_attention_8wave_bit_identical_to_4wave():
    """attn_global8_kernel (8 waves, SIMD partners in anti-phase, LDS-DMA staging) keeps attn_global_kernel's arithmetic per
    query for the rel-pos instances: the same inputs through both kernels give the same bits (sha256 of the whole output); the instances
    without rel-pos differ in where -m is added (matrix pipe / vector pipe) and agree within the output rounding.  The A/B switch
    WM_ATTN_4WAVE is read once per process, so each arm runs in a child process."""
    import subprocess, sys, os, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs, saved = [], []
    for four_wave in ("0", "1"):
        env = dict(os.environ, WM_ATTN_4WAVE=four_wave)
        d = tempfile.mkdtemp()
        r = subprocess.run([sys.executable, "-c", _ATTN_CHILD, root, d], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l[:4] in ("fp16", "bf16")])
        saved.append(torch.load(os.path.join(d, "norel_fp16_80.pt")))
    assert len(outs[0]) == 3
    rel_lines = [[l for l in o if " True " in l] for o in outs]
    assert len(rel_lines[0]) == 2 and rel_lines[0] == rel_lines[1]                # rel-pos instances: the same bits
    assert G.rel_l2(saved[0].float(), saved[1].float()) < 2e-4                      # no rel-pos: within the fp16 output rounding


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_mha16_hfc_shape(prec):
    """HFC cross-attention geometry: 8 heads x 128, q from one buffer, k/v interleaved in another."""
    B, heads, hd, n = 1, 8, 128, 4096
    dev = G.dev()
    q = G.to16(torch.randn(B * n, heads * hd, device=dev), prec)
    kv = G.to16(torch.randn(B * n, 2 * heads * hd, device=dev), prec)
    from wildlifemapper_amd import _native as Nn
    out = torch.empty((B * n, heads * hd), device=dev, dtype=G.PRECS[prec][1])
    Nn.check(Nn.lib().wm_op_mha16(Nn.ptr(q), heads * hd, Nn.ptr(kv), 2 * heads * hd, C_off(kv, heads * hd), 2 * heads * hd,
                                  Nn.ptr(out), heads * hd, B, heads, hd, n, n, G.PRECS[prec][0], G.sp()))
    qf = q.float().view(B, n, heads, hd).permute(0, 2, 1, 3)
    kf = kv.float()[:, :heads * hd].reshape(B, n, heads, hd).permute(0, 2, 1, 3)
    vf = kv.float()[:, heads * hd:].reshape(B, n, heads, hd).permute(0, 2, 1, 3)
    p = _rnd(((qf @ kf.transpose(-1, -2)) / math.sqrt(hd)).softmax(-1), prec)
    ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * n, heads * hd)
    assert G.rel_l2(out.float(), ref) < 2 * OUT16_TOL[prec]


@pytest.mark.parametrize("nq,nk,hd", [(256, 128, 80), (512, 192, 80), (256, 256, 64), (128, 192, 80), (384, 64, 80)])
def test_mha16_short_key_counts(nq, nk, hd):
    """Key counts of 2, 3 and 4 tiles of 64: the 8-wave kernel's 3-slot K / V ring below, at and just past its depth (nq a
    multiple of 256, nk >= 128); (128, 192) and (384, 64) fall outside its geometry and take the 4-wave kernel."""
    B, heads, prec, dev = 2, 2, "fp16", G.dev()
    torch.manual_seed(nq + nk)
    q = G.to16(torch.randn(B * nq, heads * hd, device=dev), prec)
    k = G.to16(torch.randn(B * nk, heads * hd, device=dev), prec)
    v = G.to16(torch.randn(B * nk, heads * hd, device=dev), prec)
    out = G.mha16(q, k, v, B, heads, hd, nq, nk, prec)
    qf = q.float().view(B, nq, heads, hd).permute(0, 2, 1, 3)
    kf = k.float().view(B, nk, heads, hd).permute(0, 2, 1, 3)
    vf = v.float().view(B, nk, heads, hd).permute(0, 2, 1, 3)
    p = _rnd(((qf @ kf.transpose(-1, -2)) / math.sqrt(hd)).softmax(-1), prec)
    ref = (p @ vf).permute(0, 2, 1, 3).reshape(B * nq, heads * hd)
    assert G.rel_l2(out.float(), ref) < 2 * OUT16_TOL[prec]


def C_off(t, elems):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + elems * t.element_size())


@pytest.mark.parametrize("nq,nk,heads,hd", [(51, 4096, 8, 16), (4096, 51, 8, 16), (51, 51, 8, 32), (7, 130, 2, 16),
                                            (64, 1024, 2, 16), (1, 2048, 3, 16), (65, 1024, 2, 16), (51, 1100, 2, 16)])
def test_mha32(nq, nk, heads, hd):
    """(51, 4096), (64, 1024), (1, 2048): the key-split kernels (keys over workgroups + chunk merge); (65, 1024) and (51, 1100) fall
    outside their geometry (nq <= 64, nk a multiple of 256) and take the query-group kernel."""
    B, dev = 2, G.dev()
    q = torch.randn(B, nq, heads * hd, device=dev)
    k = torch.randn(B, nk, heads * hd, device=dev)
    v = torch.randn(B, nk, heads * hd, device=dev)
    out = G.mha32(q, k, v, heads)
    ref = O.mha_core(q.cpu(), k.cpu(), v.cpu(), heads, O.OracleCfg())
    assert (out.cpu() - ref).abs().max().item() < 2e-5


# ---------------------------------------------------------------------------
# fp8 (BASELINE.json configs[4]): e4m3 converts and the block-scaled-MFMA GEMM (gemm8.h)
# ---------------------------------------------------------------------------
def test_fp8_convert_matches_torch_e4m3():
    """Device f32 -> e4m3 (v_cvt_pk_fp8_f32 behind a clamp) against torch's float8_e4m3fn cast on the CPU: round to
    nearest even, subnormals kept, saturation at 448; and the decode table against torch's."""
    x = torch.randn(1 << 16, device=G.dev()) * torch.exp(torch.randn(1 << 16, device=G.dev()) * 3)
    x[:16] = torch.tensor([0.0, -0.0, 1.0625, 1.1875, 447.0, 448.0, 449.0, 1e6, -1e6, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 0.0146484375,
                           -0.017, 463.9, 2.0 ** -11], device=G.dev())
    got = G.to_fp8(x).cpu()
    want = x.cpu().clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(got, want.view(torch.uint8))
    lut = G.e4m3_lut("cpu")
    allb = torch.arange(256, dtype=torch.uint8)
    ref = allb.view(torch.float8_e4m3fn).float()
    ok = ~torch.isnan(ref)
    assert torch.equal(lut[ok], ref[ok]) and bool(torch.isnan(lut[~ok]).all())


def test_gemm8_identity_asymmetric():
    """A = I against an asymmetric W (exactly representable integers) catches a wrong lane map or a transposed C write."""
    dev = G.dev()
    M, N, K = 256, 256, 256
    a = torch.zeros(M, K, device=dev)
    a[:, :K] = torch.eye(K, device=dev)[:M]
    w = ((torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) * 7) % 15) - 7       # integers -7..7, asymmetric
    a8, w8 = G.to_fp8(a), G.to_fp8(w)
    one = torch.ones(N, device=dev)
    out = G.gemm8(a8, w8, one, None, torch.zeros(M, N, device=dev), 0, "f32")
    assert torch.equal(out, w[:, :M].t().contiguous())


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 1280), (16384, 1280, 1280), (16384, 3840, 1280), (4096, 1280, 5120)])
def test_gemm8_residual_fp32(M, N, K):
    dev = G.dev()
    a8 = G.to_fp8(torch.randn(M, K, device=dev))
    w8, sc = G.quant_weight_fp8(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev) * 2 + torch.arange(M, device=dev).view(M, 1) % 61 * 0.5
    out, var = _variants_run(lambda: G.gemm8(a8, w8, sc, bias, res, 0, "f32"))
    assert var == {"fp8_256": 1}
    want = res + (G.from_fp8(a8) @ G.from_fp8(w8).t()) * sc + bias
    assert G.rel_l2(out, want) < 1e-5
    assert (out - want).abs().max().item() < 2e-3


@pytest.mark.parametrize("M,N,K,act", [(256, 256, 256, 0), (16384, 3840, 1280, 0), (16384, 5120, 1280, 1), (512, 1024, 768, 2)])
def test_gemm8_16bit_and_fp8_outputs(M, N, K, act):
    dev = G.dev()
    a8 = G.to_fp8(torch.randn(M, K, device=dev))
    w8, sc = G.quant_weight_fp8(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = torch.randn(N, device=dev)
    y = (G.from_fp8(a8) @ G.from_fp8(w8).t()) * sc + bias
    y = {0: y, 1: O.gelu_erf(y), 2: torch.relu(y)}[act]
    o16 = G.gemm8(a8, w8, sc, bias, None, act, "16", "bf16")
    assert G.rel_l2(o16.float(), y) < OUT16_TOL["bf16"]
    o8 = G.gemm8(a8, w8, sc, bias, None, act, "8")
    got = G.from_fp8(o8)
    # e4m3 output: one rounding to 3 mantissa bits (relative 2^-4 worst case, 0.036 rms) of the fp32 value.  Its GELU is the
    # logistic form (wm_common.h gelu_e4m3_fast2: within 2.7e-4 of the erf form, 1.8 % of the e4m3 outputs one step away): the
    # bits are checked against that form, the distance against the exact one
    y8 = O.gelu_logistic((G.from_fp8(a8) @ G.from_fp8(w8).t()) * sc + bias) if act == 1 else y
    want8 = G.from_fp8(G.to_fp8(y8))
    mism = (got != want8).float().mean().item()
    assert mism < 2e-3, mism                              # ties / fp32 summation-order differences at rounding boundaries only
    assert G.rel_l2(got, y) < 0.04


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (16384, 1280, 1280), (4096, 1280, 5120), (512, 768, 1280)])
def test_gemm8_planes_equals_fp32_residual_form(M, N, K, prec):
    """The fp8 blocks' residual GEMM on row-major planes: the same accumulators and the same fp32 sum as the fp32-residual instance fed
    float(hi) + float(lo), written back as hi' = round16(v), lo' = fp16(v - hi'): bit for bit."""
    dev = G.dev()
    dt = G.PRECS[prec][1]
    a8 = G.to_fp8(torch.randn(M, K, device=dev))
    w8, sc = G.quant_weight_fp8(torch.randn(N, K, device=dev) / math.sqrt(K))
    bias = torch.randn(N, device=dev)
    x = torch.randn(M, N, device=dev) * 2 + torch.arange(M, device=dev).view(M, 1) % 61 * 0.5
    hi, lo = G.stream_rows_split(x, prec)
    pos = G.plane_pos(N, dev)                               # column c of a row sits at pos[c]
    assert torch.equal(hi[:, pos], x.to(dt)) and torch.equal(lo[:, pos], (x - x.to(dt).float()).to(torch.float16))
    xr = G.stream_rows_merge(hi, lo, prec)
    assert torch.equal(xr, hi[:, pos].float() + lo[:, pos].float())
    assert bool(((xr - x).abs() <= x.abs() * 2.0 ** -19 + 2.0 ** -24).all())
    (hi2, lo2), var = _variants_run(lambda: G.gemm8_planes(a8, w8, sc, bias, hi, lo, prec))
    assert var == {"fp8_256_planes": 1}
    v = G.gemm8(a8, w8, sc, bias, xr, 0, "f32")
    assert torch.equal(hi2[:, pos], v.to(dt))
    assert torch.equal(lo2[:, pos], (v - v.to(dt).float()).to(torch.float16))


@pytest.mark.parametrize("C", [1280, 1024, 768])
def test_layernorm_fp8_on_the_hi_plane(C):
    """LayerNorm to e4m3 read from the 16-bit hi plane and written in plane order: against e4m3 of torch's LayerNorm of float(hi) (same
    values up to rounding-boundary cases of the 3-bit mantissa) and against the column-tiled fp32-input kernel; and a K-permuted weight
    undoes the order in the GEMM that consumes it."""
    dev = G.dev()
    x = torch.randn(4096, C, device=dev) * 3 + 0.5
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    pos = G.plane_pos(C, dev)
    for prec in ("bf16", "fp16"):
        hi, _ = G.stream_rows_split(x, prec)
        xh = hi[:, pos].float().contiguous()
        got = G.layernorm_fp8_plane(hi, g, b, 1e-6, prec)[:, pos].contiguous()       # back to column order
        ref = torch.nn.functional.layer_norm(xh, (C,), g, b, 1e-6)
        assert (G.from_fp8(got) != G.from_fp8(G.to_fp8(ref))).float().mean().item() < 2e-3
        assert (got != G.layernorm_fp8(xh, g, b, 1e-6)).float().mean().item() < 2e-3
        assert G.rel_l2(G.from_fp8(got), ref) < 0.04
    # the consumer: A in plane order x W with K permuted alike == A in column order x W  (exact products, fp32 sums in another order)
    w8, sc = G.quant_weight_fp8(torch.randn(256, C, device=dev) / math.sqrt(C))
    a_plane = G.layernorm_fp8_plane(hi, g, b, 1e-6, prec)
    wk = torch.empty_like(w8)
    wk[:, pos] = w8
    y_plane = G.gemm8(a_plane, wk, sc, None, None, 0, "16", "fp16").float()
    y_cols = G.gemm8(a_plane[:, pos].contiguous(), w8, sc, None, None, 0, "16", "fp16").float()
    assert G.rel_l2(y_plane, y_cols) < 1e-3


def test_gemm8_rejects_bad_shapes():
    dev = G.dev()
    a8 = torch.zeros(256, 128, device=dev, dtype=torch.uint8)
    w8 = torch.zeros(256, 128, device=dev, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="gemm8"):
        G.gemm8(a8, w8, torch.ones(256, device=dev), None, None, 0, "16")
