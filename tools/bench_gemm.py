"""Time every instantiated tile of mf_gemm_conv on a few representative shapes (3x3 conv, bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

hip.AUTOTUNE = False
dev = "cuda"
prec = ops.Precision.get("bf16")
lib = hip.load()
shapes = [  # (B, H, W, Cin, Cout, k)
    (8, 64, 64, 320, 320, 3), (8, 64, 64, 640, 320, 3), (8, 64, 64, 960, 320, 3), (8, 32, 32, 640, 640, 3),
    (8, 32, 32, 1280, 640, 3), (8, 16, 16, 1280, 1280, 3), (8, 16, 16, 2560, 1280, 3),
    (8, 64, 64, 320, 320, 1), (8, 16, 16, 1280, 1280, 1), (8, 64, 64, 320, 2560, 1), (8, 32, 32, 2560, 640, 1),
]
names = []
import ctypes
for t in range(1, lib.mf_gemm_num_tiles() + 1):
    bm, bn = ctypes.c_int(), ctypes.c_int()
    lib.mf_gemm_tile_shape(t, ctypes.byref(bm), ctypes.byref(bn))
    names.append(f"{bm.value}x{bn.value}")
print("tiles:", " ".join(f"{i + 1}:{n}" for i, n in enumerate(names)), "(7-12 = 3-stage ring)")
for (b, h, w, ci, co, k) in shapes:
    x = torch.randn(b, h, w, ci, device=dev).bfloat16()
    wt = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, dev)
    flops = 2.0 * b * h * w * co * ci * k * k
    res = []
    for t in range(1, len(names) + 1):
        try:
            for _ in range(2):
                ops.conv2d(x, wt, padding=k // 2, tile=t, splitk=1)
        except hip.MfhipError:
            res.append("    -")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(x, wt, padding=k // 2, tile=t, splitk=1)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        res.append(f"{flops / us / 1e6:5.0f}")
    print(f"M={b * h * w:6d} N={co:5d} K={ci * k * k:6d}: TF/s per tile " + " ".join(res))
