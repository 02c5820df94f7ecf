"""Host-side owner of one native handle (include/wm_hip.h) shared by the drop-in
nn.Modules.  Pure plumbing: parameter tensors in, device pointers through the
C-ABI, tensors out.  All arithmetic happens in libwm_hip.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from .synth import MODEL_DIMS, ModelDims


def default_precision() -> str:
    """fp16 operands by default: same MFMA rate as bf16, and the mode that meets north_star's 1e-3 on the logits on every
    weight set tried (1.7-2.4e-4; bf16 operands: 7.2e-4 .. 2.0e-3 depending on the weights).
    WM_PRECISION / args.wm_precision select bf16 or fp8."""
    return os.environ.get("WM_PRECISION", "fp16").lower()


class EngineHub:
    """Lazily creates / refreshes a wm_handle for the modules registered with it."""

    def __init__(self, embed_dim: int, depth: int, num_heads: int, global_attn_indexes: Iterable[int],
                 precision: Optional[str] = None, max_batch: Optional[int] = None) -> None:
        self.embed_dim, self.depth, self.num_heads = int(embed_dim), int(depth), int(num_heads)
        self.global_attn_indexes = tuple(int(i) for i in global_attn_indexes)
        self.precision = (precision or default_precision()).lower()
        if self.precision not in N.PREC_BY_NAME:
            raise ValueError(f"unknown precision {self.precision!r} (bf16 | fp16 | fp8)")
        self.max_batch = int(max_batch or os.environ.get("WM_MAX_BATCH", 0) or 0)
        # folded LayerNorm (wm_config.flags & WM_CFG_FOLD_LN): True (default; WM_LN_FOLD=0 turns it off) = the fp16-operand blocks,
        # "all" (WM_LN_FOLD=2) = bf16-operand blocks too (faster, but a different draw of bf16's ~1e-3 logits error: include/wm_hip.h);
        # set before the first forward
        self.fold_ln = {"0": False, "2": "all"}.get(os.environ.get("WM_LN_FOLD", "1"), True)
        self.fp8_gemms = 0                                         # wm_config.fp8_gemms (0 = library default); set_fp8_gemms()
        self._watch = []                                           # (owner dict, key, tensor, signature): fast no-change check
        self._handle: Optional[C.c_void_p] = None
        self._device: Optional[torch.device] = None
        self._sources: Dict[str, torch.nn.Module] = {}      # prefix -> module
        self._signature: Dict[str, Tuple[int, int]] = {}    # name -> (data_ptr, version)

    # -- registration -------------------------------------------------------
    def register(self, prefix: str, module: torch.nn.Module) -> None:
        self._sources[prefix] = module
        self._watch = []                                    # a module added after a forward: full walk at the next call

    def adopt(self, other: "EngineHub") -> None:
        """Merge the modules of another hub into this one (MedSAM wiring)."""
        if other is self:
            return
        for prefix, mod in other._sources.items():
            self._sources[prefix] = mod
            mod._hub = self
        self._watch = []
        other.close()

    # -- handle lifetime ----------------------------------------------------
    def close(self) -> None:
        if self._handle is not None:
            N.lib().wm_destroy(self._handle)
            self._handle = None
            self._signature = {}
            self._watch = []

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_precision(self, precision: str) -> None:
        precision = precision.lower()
        if precision not in N.PREC_BY_NAME:
            raise ValueError(precision)
        if precision != self.precision:
            self.precision = precision
            self.close()

    def set_fp8_gemms(self, mask: int) -> None:
        """fp8 mode: which of a block's GEMMs run e4m3 (N.FP8_QKV | N.FP8_PROJ | N.FP8_MLP; 0 = the library default)."""
        mask = int(mask) & N.FP8_ALL
        if mask != self.fp8_gemms:
            self.fp8_gemms = mask
            self.close()

    def invalidate(self) -> None:
        """Force a full weight walk at the next call.  Needed after writes that bypass autograd's version counter
        (`p.data.copy_()`, `p.data.add_()`: `p._version` does not move).  Noticed without it: a replaced Parameter / buffer,
        an in-place op on the Parameter itself (`p.mul_()`, `load_state_dict`), a swapped sub-module, a module registered
        or adopted after a forward."""
        self._watch = []

    def _create(self, device: torch.device, batch: int) -> None:
        self.close()
        cfg = N.WmConfig()
        cfg.embed_dim, cfg.depth, cfg.num_heads = self.embed_dim, self.depth, self.num_heads
        cfg.num_global = len(self.global_attn_indexes)
        for i, g in enumerate(self.global_attn_indexes):
            cfg.global_attn_indexes[i] = g
        self.max_batch = max(self.max_batch, batch)
        cfg.max_batch = self.max_batch
        cfg.precision = N.PREC_BY_NAME[self.precision]
        cfg.flags = (N.CFG_FOLD_LN if self.fold_ln else 0) | (N.CFG_FOLD_LN_BF16 if self.fold_ln == "all" else 0)
        cfg.fp8_gemms = self.fp8_gemms
        h = C.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        N.check(N.lib().wm_create(C.byref(cfg), idx, C.byref(h)))
        self._handle, self._device = h, device

    def _named_tensors(self):
        for prefix, mod in self._sources.items():
            for k, v in mod.state_dict(keep_vars=True).items():
                yield prefix + k, v

    def _unchanged(self) -> bool:
        """No registered parameter / buffer was replaced, moved or written since the last full walk (0.17 ms for ViT-H's
        575 tensors instead of the 1.8 ms state_dict walk, which at ViT-B / B = 1 is visible in a step)."""
        if not self._watch:
            return False
        for owner, key, t, sig in self._watch:
            if owner.get(key) is not t or (sig is not None and (t.data_ptr(), t._version) != sig):
                return False
        return True

    def _rebuild_watch(self) -> None:
        self._watch = []
        for mod in self._sources.values():
            for sub in mod.modules():
                for owner in (sub._parameters, sub._buffers):
                    for key, t in owner.items():
                        if t is not None:
                            self._watch.append((owner, key, t, (t.data_ptr(), t._version)))
                # the identity of every child: a swapped sub-module (`blocks[i] = new`) leaves the OLD module's dicts
                # unchanged, so the parent's slot is watched too (signature None = identity only)
                for key, child in sub._modules.items():
                    if child is not None:
                        self._watch.append((sub._modules, key, child, None))

    def _sync_weights(self) -> None:
        if self._unchanged():
            return
        lib = N.lib()
        changed = False
        for name, t in self._named_tensors():
            sig = (t.data_ptr(), t._version)
            if self._signature.get(name) == sig:
                continue
            host = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (C.c_int64 * host.dim())(*host.shape)
            N.check(lib.wm_load_weight(self._handle, name.encode(), C.c_void_p(host.data_ptr()), shape, host.dim()))
            self._signature[name] = sig
            changed = True
        if changed:
            N.check(lib.wm_finalize_weights(self._handle))
        self._rebuild_watch()

    def handle(self, device: torch.device, batch: int) -> C.c_void_p:
        if device.type != "cuda":
            raise RuntimeError("wildlifemapper_amd runs on a ROCm device only; got " + str(device))
        if self._handle is None or self._device != device or batch > self.max_batch:
            self._create(device, batch)
        self._sync_weights()
        return self._handle

    # -- the path -----------------------------------------------------------
    def hfc_fft(self, x: torch.Tensor) -> torch.Tensor:
        N.require_cuda(x, "hfc_fft input")
        B = x.shape[0]
        _check_image(x, 3)
        if self._handle is None or self._device != x.device or B > self.max_batch:
            self._create(x.device, B)          # the FFT needs no weights
        out = torch.empty((B, 1, 1024, 1024), device=x.device, dtype=torch.float32)
        N.check(N.lib().wm_hfc_fft(self._handle, N.ptr(x), N.ptr(out), B, N.stream_ptr(x.device)))
        return out

    def encoder_forward(self, x: torch.Tensor, x_hfc: torch.Tensor) -> torch.Tensor:
        N.require_cuda(x, "encoder input x")
        N.require_cuda(x_hfc, "encoder input x_hfc")
        _check_image(x, 3)
        _check_image(x_hfc, 1)
        B = x.shape[0]
        h = self.handle(x.device, B)
        out = torch.empty((B, 256, 64, 64), device=x.device, dtype=torch.float32)
        self._warn_overflow()
        N.check(N.lib().wm_encoder_forward(h, N.ptr(x), N.ptr(x_hfc), N.ptr(out), B, N.stream_ptr(x.device)))
        return out

    def decoder_forward(self, emb: torch.Tensor) -> Dict[str, torch.Tensor]:
        N.require_cuda(emb, "decoder input")
        if emb.dim() != 4 or tuple(emb.shape[1:]) != (256, 64, 64):
            raise RuntimeError(f"decoder input must be (B,256,64,64), got {tuple(emb.shape)}")
        B = emb.shape[0]
        h = self.handle(emb.device, B)
        logits = torch.empty((B, N.NUM_QUERIES, N.NUM_LOGITS), device=emb.device, dtype=torch.float32)
        boxes = torch.empty((B, N.NUM_QUERIES, 4), device=emb.device, dtype=torch.float32)
        N.check(N.lib().wm_decoder_forward(h, N.ptr(emb), N.ptr(logits), N.ptr(boxes), B, N.stream_ptr(emb.device)))
        return {"pred_logits": logits, "pred_boxes": boxes}

    def forward(self, x: torch.Tensor, target_sizes: Optional[torch.Tensor] = None, want_records: bool = False):
        """fft -> encoder -> decoder (-> PostProcess + NMS records) in one native call."""
        N.require_cuda(x, "model input")
        _check_image(x, 3)
        B = x.shape[0]
        h = self.handle(x.device, B)
        logits = torch.empty((B, N.NUM_QUERIES, N.NUM_LOGITS), device=x.device, dtype=torch.float32)
        boxes = torch.empty((B, N.NUM_QUERIES, 4), device=x.device, dtype=torch.float32)
        rec = torch.empty((B, N.NUM_QUERIES, 8), device=x.device, dtype=torch.float32) if want_records else None
        ts = None
        if target_sizes is not None:
            ts = target_sizes.to(device=x.device, dtype=torch.float32).contiguous()
        self._warn_overflow()
        N.check(N.lib().wm_forward(h, N.ptr(x), N.ptr(ts), N.ptr(logits), N.ptr(boxes), N.ptr(rec), B, N.stream_ptr(x.device)))
        out = {"pred_logits": logits, "pred_boxes": boxes}
        if want_records:
            out["records"] = rec
        return out

    def stream_overflow(self, reset: bool = False) -> int:
        """Non-zero if, in a forward that has finished, a value of the residual stream reached the fp16 clamp (bit 1: |x| >= 65504)
        or an operand of the decoder's fp16-split GEMMs left fp16's range (bit 2) (wm_stream_overflow; no synchronisation)."""
        return 0 if self._handle is None else int(N.lib().wm_stream_overflow(self._handle, int(reset)))

    def _warn_overflow(self) -> None:
        v = self.stream_overflow(reset=True)
        if v:
            import warnings
            if v & 1:
                warnings.warn("wildlifemapper_amd: residual-stream values reached the fp16 clamp (|x| >= 65504) in an earlier forward; "
                              "this checkpoint needs precision='bf16' (WM_PRECISION=bf16)", RuntimeWarning, stacklevel=3)
            if v & 2:
                warnings.warn("wildlifemapper_amd: a decoder GEMM operand left fp16's range (|activation| >= 65504 or |weight| >= 1023) in an "
                              "earlier forward (its products were inf / nan: the detections of that forward are wrong); set WM_GEMM32_F32=1 for the fp32-MFMA decoder GEMMs",
                              RuntimeWarning, stacklevel=3)

    # -- taps / profiling ---------------------------------------------------
    def set_tap(self, which: int) -> None:
        N.check(N.lib().wm_set_tap(self._handle, which))

    def read_tap(self, batch: int) -> torch.Tensor:
        out = torch.empty((batch, 64, 64, self.embed_dim), device=self._device, dtype=torch.float32)
        N.check(N.lib().wm_read_tap(self._handle, N.ptr(out), batch, N.stream_ptr(self._device)))
        return out

    def saturation_enable(self, on: bool) -> None:
        """Opt-in census of clamped operand values (wm_debug_saturation_*); see saturation_read."""
        N.check(N.lib().wm_debug_saturation_enable(self._handle, int(on)))

    def saturation_read(self, reset: bool = True) -> Dict[str, int]:
        """{buffer kind: elements found AT the operand type's clamp value (fp16 65504, e4m3 448) or inf / NaN (bf16)} since the
        last reset.  Non-zero in fp16 mode = clamped operands: run that checkpoint with precision bf16."""
        arr = (C.c_int64 * len(N.SAT_NAMES))()
        N.check(N.lib().wm_debug_saturation_read(self._handle, arr, len(N.SAT_NAMES), int(reset), N.stream_ptr(self._device)))
        return {n: int(arr[i]) for i, n in enumerate(N.SAT_NAMES)}

    def profile_enable(self, on: bool) -> None:
        N.check(N.lib().wm_profile_enable(self._handle, int(on)))

    def profile_reset(self) -> None:
        N.check(N.lib().wm_profile_reset(self._handle))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        arr = (N.WmKclassStat * len(N.KCLASS_NAMES))()
        N.check(N.lib().wm_profile_read(self._handle, arr))
        return {n: {"launches": int(arr[i].launches), "ms": float(arr[i].ms), "flops": float(arr[i].flops),
                    "bytes": float(arr[i].bytes)} for i, n in enumerate(N.KCLASS_NAMES)}


def postprocess_nms(logits: torch.Tensor, boxes: torch.Tensor, target_sizes: torch.Tensor,
                    conf_thr: float = 0.05, score_thr: float = 0.5, iou_thr: float = 0.4) -> torch.Tensor:
    """PostProcess + score cut + NMS kernel (wm_postprocess_nms).  Weightless: no handle, no workspace.
    Returns raw records (B,51,8) float32-viewed: x0,y0,x1,y1,score | int32 label,flags,nms_rank."""
    N.require_cuda(logits, "pred_logits")
    N.require_cuda(boxes, "pred_boxes")
    B = logits.shape[0]
    if tuple(logits.shape[1:]) != (N.NUM_QUERIES, N.NUM_LOGITS) or tuple(boxes.shape) != (B, N.NUM_QUERIES, 4):
        raise RuntimeError(f"postprocess: expected (B,51,8) logits and (B,51,4) boxes, got {tuple(logits.shape)}, {tuple(boxes.shape)}")
    ts = target_sizes.to(device=logits.device, dtype=torch.float32).contiguous()
    rec = torch.empty((B, N.NUM_QUERIES, 8), device=logits.device, dtype=torch.float32)
    with torch.cuda.device(logits.device):
        N.check(N.lib().wm_postprocess_nms(None, N.ptr(logits), N.ptr(boxes), N.ptr(ts), conf_thr, score_thr, iou_thr,
                                           N.ptr(rec), B, N.stream_ptr(logits.device)))
    return rec


def _check_image(x: torch.Tensor, chans: int) -> None:
    if x.dim() != 4 or x.shape[1] != chans or x.shape[2] != 1024 or x.shape[3] != 1024:
        raise RuntimeError(f"expected (B,{chans},1024,1024), got {tuple(x.shape)}")


def split_records(rec: torch.Tensor) -> Dict[str, torch.Tensor]:
    """(B,51,8) raw records -> named views (int fields reinterpreted)."""
    ints = rec.view(torch.int32)
    return {"boxes": rec[..., 0:4], "scores": rec[..., 4], "labels": ints[..., 5].to(torch.int64),
            "flags": ints[..., 6], "nms_rank": ints[..., 7]}


def hub_for(model_type: str, precision: Optional[str] = None, max_batch: Optional[int] = None) -> EngineHub:
    d: ModelDims = MODEL_DIMS[model_type]
    return EngineHub(d.embed_dim, d.depth, d.num_heads, d.global_attn_indexes, precision, max_batch)
