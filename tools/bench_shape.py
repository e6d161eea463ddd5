"""Graph-replayed timing of one 3x3 conv shape over (tile, split-K) candidates: bench_shape.py B H W Cin Cout"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
b, h, w, ci, co = [int(v) for v in sys.argv[1:6]]
x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
wt = ops.ConvWeight(torch.randn(co, ci, 3, 3) * 0.02, torch.randn(co), prec, "cuda")
flops = 2.0 * b * h * w * co * ci * 9
res = []
for t in (1, 14, 5, 11, 8, 13, 20, 21, 22, 23, 24, 16, 18):
    for sk in (1, 2, 3, 4, 6):
        try:
            us = timed(lambda: ops.conv2d(x, wt, padding=1, tile=t, splitk=sk))
        except hip.MfhipError:
            continue
        res.append((us, t, sk))
res.sort()
print(f"M={b*h*w} N={co} K={ci*9}: " + " | ".join(f"t{t}/sk{sk} {us:.1f}us {flops/us/1e6:.0f}TF" for us, t, sk in res[:8]), flush=True)
