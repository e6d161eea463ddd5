"""Warp-specialised plain ring (tiles 41-46) against tiles 14 / 26 / 1 / 25 on the step's 1x1 GEMMs and upsampled convs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
TILES = (14, 1, 29, 6, 41, 43, 44, 48)
for (b, h, w, ci, co, k, up) in [(8, 64, 64, 1280, 320, 1, 0), (8, 64, 64, 320, 320, 1, 0), (8, 32, 32, 2560, 640, 1, 0), (8, 32, 32, 640, 640, 1, 0),
                                  (8, 16, 16, 5120, 1280, 1, 0), (8, 16, 16, 1280, 1280, 1, 0), (8, 64, 64, 320, 1280, 1, 0),
                                  (8, 32, 32, 640, 640, 3, 1), (8, 16, 16, 1280, 1280, 3, 1), (8, 64, 64, 320, 640, 1, 0)]:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    cw = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
    ho = h * 2 if up else h
    res = torch.randn(b, ho, ho, co, device="cuda").bfloat16()
    row = []
    for tile in TILES:
        t = timed(lambda: ops.conv2d(x, cw, padding=k // 2, upsample=bool(up), tile=tile, splitk=1, res0=res))
        row.append(f"t{tile} {t:6.1f}")
    fl = 2.0 * b * ho * ho * ci * co * k * k
    print(f"B{b} {h}x{w} {ci}->{co} k{k} up{up}: " + "  ".join(row) + f"   (1000 TF/s = {fl / 1e9:.1f} us)", flush=True)
