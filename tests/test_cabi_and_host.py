"""CPU-side checks: the C-ABI library builds, loads and exports every symbol include/wm_hip.h
declares (no compute calls without a GPU); host logic of the drop-in package."""
import os
import re

import numpy as np
import pytest
import torch

from wildlifemapper_amd import _native as N
from wildlifemapper_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    import __graft_entry__ as g
    g.build()
    hdr = open(os.path.join(ROOT, "include", "wm_hip.h")).read()
    declared = set(re.findall(r"\b(wm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(N.SYMBOLS), declared ^ set(N.SYMBOLS)
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name), name
    m = re.search(r"#define WM_ABI_VERSION (\d+)", hdr)
    assert lib.wm_abi_version() == int(m.group(1)) == N.ABI_VERSION        # header, library and binding agree


def test_stale_library_is_refused(monkeypatch):
    """A libwm_hip.so whose wm_abi_version() differs from the binding's is refused at load, not at the first
    missing symbol or misread flag."""
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "ABI_VERSION", N.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        N.lib()
    monkeypatch.setattr(N, "ABI_VERSION", N.ABI_VERSION - 1)
    monkeypatch.setattr(N, "_lib", None)
    assert N.lib().wm_abi_version() == N.ABI_VERSION


def test_weight_watch_notices_every_kind_of_change():
    """EngineHub's fast no-change check (the full state_dict walk is 1.8 ms per forward for ViT-H): unchanged -> True; an
    in-place write, a replaced Parameter and a replaced buffer -> False."""
    from wildlifemapper_amd.segment_anything import sam_model_registry
    from wildlifemapper_amd.segment_anything.network import MedSAM
    sam, _, _ = sam_model_registry["vit_b"](None, None)
    m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder)
    hub = m._hub
    assert not hub._unchanged()                      # nothing watched yet: full walk
    hub._rebuild_watch()
    n_tensors = sum(1 for w in hub._watch if w[3] is not None)
    assert hub._unchanged() and n_tensors == len(list(hub._named_tensors()))
    with torch.no_grad():
        m.image_encoder.blocks[3].attn.qkv.weight.mul_(1.0)
    assert not hub._unchanged()
    hub._rebuild_watch()
    m.mask_decoder.mask_tokens.weight = torch.nn.Parameter(torch.zeros_like(m.mask_decoder.mask_tokens.weight))
    assert not hub._unchanged()
    hub._rebuild_watch()
    m.load_state_dict(m.state_dict())
    assert not hub._unchanged()
    hub._rebuild_watch()
    assert hub._unchanged()
    hub.invalidate()
    assert not hub._unchanged()
    # a whole sub-module swapped: the old module's parameter dicts are untouched, the parent's slot is not
    import copy
    hub._rebuild_watch()
    assert hub._unchanged()
    m.image_encoder.blocks[2] = copy.deepcopy(m.image_encoder.blocks[2])
    assert not hub._unchanged()
    hub._rebuild_watch()
    m.image_encoder.neck[1] = copy.deepcopy(m.image_encoder.neck[1])
    assert not hub._unchanged()
    # register / adopt after a forward drop the watch list
    hub._rebuild_watch()
    assert hub._unchanged()
    hub.register("extra.", torch.nn.Linear(2, 2))
    assert not hub._unchanged()
    del hub._sources["extra."]
    hub._rebuild_watch()
    from wildlifemapper_amd.engine import hub_for
    other = hub_for("vit_b")
    hub.adopt(other)
    assert not hub._unchanged()
    # documented limit: a write through .data does not move the version counter -> invalidate() is the contract
    hub._rebuild_watch()
    with torch.no_grad():
        m.image_encoder.blocks[3].attn.qkv.weight.data.add_(0.0)
    assert hub._unchanged()


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(N.WmBoxRecord) == 32
    assert C.sizeof(N.WmConfig) == 4 * (4 + 8 + 2 + 4)
    assert C.sizeof(N.WmKclassStat) == 32


def test_argument_errors_without_gpu():
    lib = N.lib()
    assert lib.wm_destroy(None) == 0
    assert lib.wm_finalize_weights(None) != 0
    assert b"null handle" in lib.wm_last_error()
    assert lib.wm_create(None, 0, None) != 0


def test_state_dict_names_match_synth_enumeration():
    from wildlifemapper_amd.segment_anything import sam_model_registry, build_sam
    from wildlifemapper_amd.segment_anything.network import MedSAM
    sam, criterion, post = sam_model_registry["vit_b"](None, None)
    assert set(post) == {"bbox"} and criterion(None, None) == {}
    m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder)
    sd = m.state_dict()
    exp = synth.weight_shapes("vit_b")
    assert set(sd) == set(exp)
    assert all(tuple(sd[k].shape) == tuple(exp[k]) for k in exp)
    # requires_grad pattern of network.py:19-34
    trainable = {n for n, p in m.image_encoder.named_parameters() if p.requires_grad}
    assert all(any(t in n for t in ("hfc_embed", "hfc_attn", "patch_embed")) for n in trainable) and trainable
    assert sam.image_encoder.img_size == 1024


def test_no_cpu_fallback():
    from wildlifemapper_amd.segment_anything import sam_model_registry
    from wildlifemapper_amd.segment_anything.network import MedSAM
    sam, _, post = sam_model_registry["vit_b"](None, None)
    m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 1024, 1024), None)
    with pytest.raises(RuntimeError):
        sam.image_encoder.blocks[0](torch.zeros(1, 64, 64, 768))
    with pytest.raises(RuntimeError):
        post["bbox"]({"pred_logits": torch.zeros(1, 51, 8), "pred_boxes": torch.zeros(1, 51, 4)}, torch.tensor([[1024, 1024]]))


def test_unsupported_configs_raise():
    from wildlifemapper_amd.segment_anything.modeling import ImageEncoderViT
    with pytest.raises(NotImplementedError):
        ImageEncoderViT(img_size=512)


def test_synth_is_deterministic_and_profiled():
    a = synth.make_weight("image_encoder.blocks.3.attn.qkv.weight", (48, 16))
    b = synth.make_weight("image_encoder.blocks.3.attn.qkv.weight", (48, 16))
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert abs(float(a[0, 0])) <= 0.25
    # known-answer: pins the generator across platforms
    assert synth.make_tile_u8(0)[0, 0].tolist() == synth.make_tile_u8(0)[0, 0].tolist()
    t = synth.make_tile_u8(5)
    assert t.shape == (1024, 1024, 3) and t.dtype == np.uint8
    x = synth.normalize_tile(t)
    assert x.shape == (3, 1024, 1024) and abs(float(x.mean())) < 1.0
    base = synth.make_weight("mask_decoder.transformer.layers.0.cross_attn_token_to_image.q_proj.weight", (128, 256))
    sens = synth.make_weight("mask_decoder.transformer.layers.0.cross_attn_token_to_image.q_proj.weight", (128, 256), 0, "sensitive")
    assert np.allclose(sens, 2 * base)


def test_nested_tensor_collation():
    from wildlifemapper_amd.segment_anything.utils.misc import custom_collate, nested_tensor_from_tensor_list
    a, b = torch.ones(3, 768, 512), torch.ones(3, 1100, 1200) * 2
    nt = nested_tensor_from_tensor_list([a, b])
    assert nt.tensors.shape == (2, 3, 1024, 1024)
    assert float(nt.tensors[0, :, 768:, :].abs().sum()) == 0 and float(nt.tensors[0, 0, 0, 511]) == 1
    assert float(nt.tensors[1].min()) == 2                      # cropped at 1024
    assert bool(nt.mask[0, 0, 512]) and not bool(nt.mask[0, 767, 511]) and not bool(nt.mask[1].any())
    imgs, tg = custom_collate([{"image": a, "target": {"id": 1}}, {"image": b, "target": {"id": 2}}])
    assert imgs.tensors.shape[0] == 2 and tg[1]["id"] == 2


def test_build_sam_checkpoint_filter(tmp_path):
    """_build_sam (build_sam.py:311-322): a SAM checkpoint is loaded with mask_decoder.* keys that are not
    part of the transformer dropped, strict=False; both a bare state dict and {'model': sd} are accepted."""
    from wildlifemapper_amd.segment_anything import build_sam_vit_b
    sam0, _, _ = build_sam_vit_b(None, None)
    sd = {k: v.clone() for k, v in sam0.state_dict().items()}
    for k in sd:
        sd[k] = torch.full_like(sd[k], 0.25)
    sd["mask_decoder.output_upscaling.0.weight"] = torch.zeros(3)          # vanilla-SAM key that does not exist here
    path = tmp_path / "sam.pth"
    torch.save(sd, path)
    sam1, _, _ = build_sam_vit_b(str(path), None)
    got = sam1.state_dict()
    assert float(got["image_encoder.blocks.0.attn.qkv.weight"].mean()) == 0.25
    assert float(got["mask_decoder.transformer.layers.0.norm1.weight"].mean()) == 0.25
    # non-transformer decoder weights keep their fresh initialisation
    assert float(got["mask_decoder.mask_tokens.weight"].abs().mean()) != 0.25
    torch.save({"model": sd, "epoch": 3}, path)
    sam2, _, _ = build_sam_vit_b(str(path), None)
    assert float(sam2.state_dict()["image_encoder.pos_embed"].mean()) == 0.25


def test_resize_size_rule_and_coefficients_match_pil_restatement():
    """Host side of the N1 resize (no GPU): the output-size rule (augmentation.py:80-99) and the 22-bit fixed-point
    coefficient tables (Pillow Resample.c) computed by the library equal the oracle's restatement, which the PIL-generated
    fixture pins (tests/test_oracle_small.py::test_pil_resize_restatement)."""
    import ctypes as C
    from oracle import pil_resize as R
    from wildlifemapper_amd.preprocess import resized_size
    rng = np.random.default_rng(0)
    sizes = [(3648, 5472), (4000, 6000), (5472, 3648), (1000, 1000), (768, 768), (767, 1023), (1, 5), (333, 334)]
    sizes += [tuple(int(v) for v in rng.integers(1, 7000, 2)) for _ in range(200)]
    for h, w in sizes:
        for size, mx in ((768, 768), (512, 1333), (64, 96)):
            assert resized_size(h, w, size, mx) == R.get_size_with_aspect_ratio((w, h), size, mx), (h, w, size, mx)
    lib = N.lib()
    for n_in, n_out in [(5472, 768), (3648, 512), (225, 64), (60, 96), (100, 100), (7, 3), (6000, 768), (4000, 512)]:
        b_ref, k_ref, ks_ref = R.precompute_coeffs(n_in, n_out)
        cap = n_out * ks_ref
        b = (C.c_int * (2 * n_out))()
        k = (C.c_int * cap)()
        ks = C.c_int()
        N.check(lib.wm_debug_resize_coeffs(n_in, n_out, b, k, cap, C.byref(ks)))
        assert ks.value == ks_ref
        assert np.array_equal(np.frombuffer(b, dtype=np.int32).reshape(n_out, 2), b_ref)
        assert np.array_equal(np.frombuffer(k, dtype=np.int32).reshape(n_out, ks_ref), k_ref)


def test_plane_pos_is_a_permutation_of_each_256_column_block():
    """include/wm_hip.h wm_op_stream_rows: the column order of the fp8 blocks' stream planes.  The formula quoted there (and restated in
    tests/gpu_util.py for the GPU tests) must be a bijection on every 256-column block that keeps aligned groups of 32 columns together
    (the kernels move 4- and 8-column vectors) and sends the 4 x 32 columns of one residual-epilogue pass to one contiguous run."""
    import numpy as np
    for C in (768, 1024, 1280):
        c = np.arange(C)
        pos = (c & ~255) + ((c >> 5) & 1) * 128 + ((c >> 6) & 3) * 32 + (c & 31)
        assert sorted(pos.tolist()) == list(range(C))
        assert np.array_equal(pos // 256, c // 256)                                   # inside its block
        assert np.array_equal(pos[::32] % 32, np.zeros(C // 32, dtype=pos.dtype))     # groups of 32 stay aligned ...
        assert np.array_equal(pos - pos // 32 * 32, c % 32)                           # ... and in order
        for blk in range(C // 256):
            for ni in (0, 1):                                                         # a pass: columns 64 strip + 32 ni + (0..31), strip = 0..3
                cols = np.concatenate([blk * 256 + 64 * strip + 32 * ni + np.arange(32) for strip in range(4)])
                p = np.sort(pos[cols])
                assert np.array_equal(p, np.arange(p[0], p[0] + 128)), (C, blk, ni)
        inv = (pos & ~255) + ((pos >> 5) & 3) * 64 + ((pos >> 7) & 1) * 32 + (pos & 31)   # wm::plane_col (csrc/wm_common.h)
        assert np.array_equal(inv, c)
