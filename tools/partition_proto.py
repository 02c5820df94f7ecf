#!/usr/bin/env python3
"""Dev prototype: P concurrent pipelines, each on its own CU-masked stream (hipExtStreamCreateWithCUMask: on this part a mask
must keep CUs on every XCD, so a partition = the same CU slots of all 8 XCDs) and its own handle with 16 / P tiles, against
one pipeline with 16 tiles on the whole chip.  Same kernels, same launches; only where and when they run changes."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wildlifemapper_amd import synth
from wildlifemapper_amd.segment_anything import sam_model_registry
from wildlifemapper_amd.segment_anything.network import MedSAM

ap = argparse.ArgumentParser()
ap.add_argument("--parts", default="1,2,4")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--tiles", type=int, default=16)
ap.add_argument("--nomask", action="store_true", help="plain streams instead of CU-masked ones")
ap.add_argument("--offset-us", type=float, default=0.0, help="start partition p this many microseconds x p after partition 0 (once; no cross-partition sync afterwards)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")

def masked_stream(p, P):
    s = C.c_void_p()
    if a.nomask or P == 1:
        rc = hip.hipStreamCreateWithFlags(C.byref(s), 1)          # hipStreamNonBlocking
    else:
        words = [0] * 8
        per = 8 // P                                              # 32-bit words per partition: CU slots [32 p / P .. ) of every XCD
        for w in range(p * per, (p + 1) * per):
            words[w] = 0xffffffff
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, (C.c_uint32 * 8)(*words))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)

sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("vit_h").items()}
def build():
    sam, _, _ = sam_model_registry["vit_h"](None, None)
    m = MedSAM(sam.image_encoder, sam.mask_decoder, sam.prompt_encoder).eval()
    m.load_state_dict(sd, strict=True)
    m._hub.set_precision(a.precision)
    return m

T = a.tiles
x_all = torch.from_numpy(synth.make_batch(0, T)).to(dev)
ts_all = torch.full((T, 2), 1024.0, device=dev)
ref = None
for P in [int(v) for v in a.parts.split(",")]:
    B = T // P
    models = [build() for _ in range(P)]
    streams = [masked_stream(p, P) for p in range(P)]
    xs = [x_all[p * B:(p + 1) * B].contiguous() for p in range(P)]
    tss = [ts_all[p * B:(p + 1) * B].contiguous() for p in range(P)]
    outs = [None] * P
    def run(steps):
        for _ in range(steps):
            for p in range(P):
                with torch.cuda.stream(streams[p]):
                    outs[p] = models[p].detect(xs[p], tss[p])
    with torch.no_grad():
        run(2)
        torch.cuda.synchronize()
        if a.offset_us > 0:
            for p in range(1, P):
                with torch.cuda.stream(streams[p]):
                    torch.cuda._sleep(int(a.offset_us * p * 100))      # device spin, ~100 MHz counter
        t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rec = torch.cat([o["records"] for o in outs], 0)
    if ref is None:
        ref = rec.clone()
    same = torch.equal(rec.view(torch.int32), ref.view(torch.int32))
    print(f"P={P} x B={B}{' (plain streams)' if a.nomask else ''} offset {a.offset_us:.0f} us: {T * a.steps / dt:7.2f} tiles/s  ({dt / a.steps * 1e3:.1f} ms per {T} tiles)  records identical to P=1: {same}", flush=True)
    for m in models:
        m._hub.close()
    del models
    torch.cuda.empty_cache()
