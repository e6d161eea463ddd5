"""Split-K sweep for the 32x32-level 3x3 convs (one block per CU at split-K 1): tile x split-K, graph-replayed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
for (b, h, w, ci, co) in [(8, 32, 32, 640, 640), (8, 32, 32, 1280, 640), (8, 32, 32, 960, 640), (8, 32, 32, 320, 640), (8, 16, 16, 1280, 1280)]:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    cw = ops.ConvWeight(torch.randn(co, ci, 3, 3) * 0.02, torch.randn(co), prec, "cuda")
    res = torch.randn(b, h, w, co, device="cuda").bfloat16()
    for tile in (20, 27, 21, 22, 1):
        row = []
        for sk in (1, 2, 3, 4):
            try:
                t = timed(lambda: ops.conv2d(x, cw, padding=1, tile=tile, splitk=sk, res0=res))
                row.append(f"sk{sk} {t:6.1f}")
            except hip.MfhipError:
                row.append(f"sk{sk}   n/a")
        fl = 2.0 * b * h * w * ci * co * 9
        print(f"B{b} {h}x{w} {ci}->{co} tile {tile:2d}: " + "  ".join(row) + f"   (1000 TF/s = {fl / 1e9:.1f} us)", flush=True)
