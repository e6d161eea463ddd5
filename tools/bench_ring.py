"""Ring-depth experiment at EQUAL occupancy: one conv / GEMM under a list of tiles (argv[1]); run with MFHIP_SMEM_MIN so
that the 2-stage and 3/4/6-stage variants of one tile shape get the same number of blocks per CU."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
tiles = [int(t) for t in sys.argv[1].split(",")]
print("MFHIP_SMEM_MIN =", os.environ.get("MFHIP_SMEM_MIN"))
for (b, h, w, ci, co, k) in [(8, 64, 64, 320, 320, 3), (8, 32, 32, 640, 640, 3), (8, 64, 64, 1280, 320, 1)]:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    cw = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
    for tile in tiles:
        t = timed(lambda: ops.conv2d(x, cw, padding=k // 2, tile=tile, splitk=1))
        fl = 2.0 * b * h * w * ci * co * k * k
        print(f"B{b} {h}x{w} {ci}->{co} k{k} tile {tile:2d}: {t:7.1f} us {fl / t / 1e6:6.0f} TF/s", flush=True)
