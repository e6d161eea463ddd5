"""Run one GEMM / conv a few times in a given precision (for rocprofv3 --pmc passes).
usage: one_gemm.py PREC B H W Cin Cout k [tile]     PREC in bf16 | f16x3 | fp8 | fp32"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

hip.AUTOTUNE = False
pn = sys.argv[1]
b, h, w, ci, co, k = [int(v) for v in sys.argv[2:8]]
tile = int(sys.argv[8]) if len(sys.argv) > 8 else 0
if pn == "fp8":
    prec = ops.Precision.get("fp8")
    lw = ops.ConvWeight(torch.randn(co, ci) * 0.02, torch.randn(co), prec, "cuda", fp8=True)
    x = hip.quantize_rows_fp8(torch.randn(b * h * w, ci, device="cuda").bfloat16())
    for _ in range(5):
        ops.linear(x, lw, tile=tile)
else:
    prec = ops.Precision.get(pn)
    x = torch.randn(b, h, w, ci, device="cuda").to(prec.act)
    wt = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
    for _ in range(5):
        ops.conv2d(x, wt, padding=k // 2, tile=tile, splitk=1)
torch.cuda.synchronize()
