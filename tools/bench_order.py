"""Tile-order experiment: the GEMM / conv shapes of the step whose W (or A) does not fit one XCD's L2, timed under the
tile order forced by MFHIP_ORD ("mfast,pw"); run once per order.  Graph-replayed (tools/bench_k.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
# (B, H, W, Cin, Cout, k, tile, splitk, geglu)
SHAPES = [(8, 16, 16, 1280, 1280, 3, 1, 3, 0), (8, 8, 8, 1280, 1280, 3, 8, 6, 0), (8, 16, 16, 2560, 1280, 3, 1, 3, 0),
          (8, 8, 8, 2560, 1280, 3, 11, 12, 0), (8, 32, 32, 640, 5120, 1, 1, 1, 1), (8, 16, 16, 1280, 10240, 1, 13, 1, 1),
          (8, 64, 64, 320, 2560, 1, 14, 1, 1), (8, 32, 32, 640, 640, 3, 20, 1, 0), (8, 64, 64, 320, 320, 3, 20, 1, 0),
          (4, 64, 64, 640, 5120, 1, 14, 1, 0)]
print("MFHIP_ORD =", os.environ.get("MFHIP_ORD"))
ws = torch.empty(64 << 20, device="cuda")
for (b, h, w, ci, co, k, tile, sk, geglu) in SHAPES:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    if k == 1:
        lw = ops.ConvWeight(torch.randn(co, ci) * 0.02, torch.randn(co), prec, "cuda")
        x2 = x.view(-1, ci)
        fn = lambda: ops.linear(x2, lw, tile=tile, splitk=sk)
    else:
        cw = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
        fn = lambda: ops.conv2d(x, cw, padding=1, tile=tile, splitk=sk)
    t = timed(fn)
    fl = 2.0 * b * h * w * ci * co * k * k
    print(f"B{b} {h}x{w} {ci}->{co} k{k} tile {tile} sk {sk}: {t:7.1f} us {fl / t / 1e6:6.0f} TF/s", flush=True)
