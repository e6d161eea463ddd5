import os, sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
for (b, h, w, ci, co) in [(8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 32, 32, 640, 640), (4, 512, 512, 256, 128),
                          (4, 256, 256, 512, 256), (4, 128, 128, 512, 512)]:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    wt = ops.ConvWeight(torch.randn(co, ci, 3, 3) * 0.02, torch.randn(co), prec, "cuda")
    flops = 2.0 * b * h * w * co * ci * 9
    r = []
    for t in (1, 14, 16, 18, 20, 21):
        us = timed(lambda: ops.conv2d(x, wt, padding=1, tile=t, splitk=1))
        r.append(f"tile{t}: {us:7.1f} us {flops / us / 1e6:5.0f} TF/s")
    print(f"M={b*h*w} N={co} K={ci*9}: " + " | ".join(r), flush=True)
