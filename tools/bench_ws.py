"""Warp-specialised dx-reuse conv (tiles 37, 38) against tiles 20 / 27 / 1, graph-replayed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
for (b, h, w, ci, co, tiles) in [(8, 64, 64, 320, 320, (27, 37, 38, 39, 40, 47)), (8, 64, 64, 640, 320, (27, 37, 38, 39, 40, 47)), (8, 64, 64, 960, 320, (27, 37, 38, 39, 40, 47)),
                                  (8, 32, 32, 640, 640, (27, 1, 37, 38, 39, 40, 47)), (8, 32, 32, 1280, 640, (27, 37, 38, 39, 40, 47)),
                                  (8, 16, 16, 1280, 1280, (27, 37, 38, 39, 40, 47))]:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    cw = ops.ConvWeight(torch.randn(co, ci, 3, 3) * 0.02, torch.randn(co), prec, "cuda")
    res = torch.randn(b, h, w, co, device="cuda").bfloat16()
    row = []
    for tile in tiles:
        for sk in ((1,) if h > 32 else (1, 2) if h == 32 else (2, 4)):
            try:
                t = timed(lambda: ops.conv2d(x, cw, padding=1, tile=tile, splitk=sk, res0=res))
                row.append(f"t{tile}/sk{sk} {t:6.1f}")
            except hip.MfhipError as e:
                row.append(f"t{tile}/sk{sk} n/a")
    fl = 2.0 * b * h * w * ci * co * 9
    print(f"B{b} {h}x{w} {ci}->{co}: " + "  ".join(row) + f"   (1000 TF/s = {fl / 1e9:.1f} us)", flush=True)
