"""A/B of mf_gemm_conv tiles on the denoise step's own shapes (bf16, batch 4 with CFG: 8 images), each launch replayed from a
hipGraph.  Every candidate's output is compared with the first candidate's (tiles differ in speed, not in results: same K order
without split-K).  Run it once per library to compare builds on one box:
    MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_r04.so python tools/bench_tiles.py --set conv --tiles 39,48,42
    python tools/bench_tiles.py --set conv --tiles 39,48,42,49,50
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402
from bench_k import timed  # noqa: E402

hip.AUTOTUNE = False

# (label, B, H, W, Cin, Cout, k, upsample, with residual)
SETS = {
    "conv": [("conv 64^2 320->320", 8, 64, 64, 320, 320, 3, False, False), ("conv 64^2 640->320", 8, 64, 64, 640, 320, 3, False, False),
             ("conv 64^2 960->320", 8, 64, 64, 960, 320, 3, False, False), ("conv 32^2 640->640", 8, 32, 32, 640, 640, 3, False, False),
             ("conv 32^2 1280->640", 8, 32, 32, 1280, 640, 3, False, False), ("conv 32^2 1920->640", 8, 32, 32, 1920, 640, 3, False, False),
             ("conv 32^2 320->640", 8, 32, 32, 320, 640, 3, False, False),
             ("conv 16^2 1280->1280", 8, 16, 16, 1280, 1280, 3, False, False), ("conv 16^2 2560->1280", 8, 16, 16, 2560, 1280, 3, False, False),
             ("conv 16^2 640->1280", 8, 16, 16, 640, 1280, 3, False, False),
             ("conv 8^2 1280->1280", 8, 8, 8, 1280, 1280, 3, False, False), ("conv 8^2 2560->1280", 8, 8, 8, 2560, 1280, 3, False, False)],
    "up": [("up 32->64 640", 8, 32, 32, 640, 640, 3, True, False), ("up 16->32 1280", 8, 16, 16, 1280, 1280, 3, True, False),
           ("up 8->16 1280", 8, 8, 8, 1280, 1280, 3, True, False)],
    "lin": [("1x1 64^2 320->320 +res", 8, 64, 64, 320, 320, 1, False, True), ("1x1 32^2 640->640 +res", 8, 32, 32, 640, 640, 1, False, True),
            ("1x1 16^2 1280->1280 +res", 8, 16, 16, 1280, 1280, 1, False, True), ("1x1 8^2 1280->1280 +res", 8, 8, 8, 1280, 1280, 1, False, True),
            ("1x1 64^2 1280->320 +res", 8, 64, 64, 1280, 320, 1, False, True), ("1x1 32^2 2560->640 +res", 8, 32, 32, 2560, 640, 1, False, True),
            ("1x1 16^2 5120->1280 +res", 8, 16, 16, 5120, 1280, 1, False, True), ("1x1 64^2 640->320", 8, 64, 64, 640, 320, 1, False, False),
            ("1x1 32^2 1280->640", 8, 32, 32, 1280, 640, 1, False, False), ("1x1 16^2 2560->1280", 8, 16, 16, 2560, 1280, 1, False, False),
            ("1x1 64^2 320->2560", 8, 64, 64, 320, 2560, 1, False, False), ("1x1 32^2 640->5120", 8, 32, 32, 640, 5120, 1, False, False),
            ("1x1 16^2 1280->10240", 8, 16, 16, 1280, 10240, 1, False, False)],
}

ap = argparse.ArgumentParser()
ap.add_argument("--set", default="conv,up,lin")
ap.add_argument("--tiles", default="", help="comma list of tile[:splitk] candidates (besides the tuned one of the shipped cache)")
ap.add_argument("--only", default="", help="substring filter on the case label")
ap.add_argument("--prec", default="bf16", help="precision mode: bf16, fp16, f16x3, fp32")
a = ap.parse_args()
prec = ops.Precision.get(a.prec)
extra = []
for t in filter(None, a.tiles.split(",")):
    tt, _, sk = t.partition(":")
    extra.append((int(tt), int(sk) if sk else 0))

print(f"library: {os.environ.get('MFHIP_LIB') or 'product build'}; precision {a.prec}", flush=True)
for name in a.set.split(","):
    for (label, b, h, w, ci, co, k, ups, with_res) in SETS[name]:
        if a.only and a.only not in label:
            continue
        x = torch.randn(b, h, w, ci, device="cuda").to(prec.act)
        cw = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
        ho, wo = (2 * h, 2 * w) if ups else (h, w)
        res = torch.randn(b, ho, wo, co, device="cuda").to(prec.act) if with_res else None
        fl = 2.0 * b * ho * wo * ci * co * k * k
        # the tuned (tile, split-K) of the shipped cache for this call
        hip.AUTOTUNE = True
        probe = {}
        orig = hip._tuned_config
        def spy(d, key, _o=orig, _p=probe):
            r = _o(d, key)
            _p["cfg"] = r
            return r
        hip._tuned_config = spy
        try:
            ops.conv2d(x, cw, padding=k // 2, upsample=ups, res0=res)
        finally:
            hip._tuned_config = orig
            hip.AUTOTUNE = False
        tuned = probe.get("cfg", (0, 0))
        cands = [(int(tuned[0]), int(tuned[1]))] + [c for c in extra if c != (int(tuned[0]), int(tuned[1]))]
        ref = None
        row = []
        for (tile, sk) in cands:
            sks = [sk] if sk else ([1] if b * ho * wo >= 8192 else [1, 2, 4])
            for s in sks:
                try:
                    y = ops.conv2d(x, cw, padding=k // 2, upsample=ups, res0=res, tile=tile, splitk=s)
                    t = timed(lambda: ops.conv2d(x, cw, padding=k // 2, upsample=ups, res0=res, tile=tile, splitk=s))
                except hip.MfhipError:
                    continue
                if ref is None:
                    ref = y.float()
                    err = 0.0
                else:
                    err = float((y.float() - ref).abs().max())
                flag = "" if err <= 0.02 * float(ref.abs().max()) else f" DIFF {err:.3g}"
                row.append(f"t{tile}/sk{s} {t:6.1f}{flag}")
        print(f"{label:26s} ({fl / 1e9:6.1f} us at 1 PF/s) tuned " + "  ".join(row), flush=True)
