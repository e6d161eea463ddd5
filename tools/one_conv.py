"""Run one conv shape a few times (for rocprofv3 --pmc passes). usage: one_conv.py B H W Cin Cout k tile"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

hip.AUTOTUNE = False
b, h, w, ci, co, k, tile = [int(v) for v in sys.argv[1:8]]
prec = ops.Precision.get("bf16")
x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
wt = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
for _ in range(5):
    ops.conv2d(x, wt, padding=k // 2, tile=tile, splitk=1)
torch.cuda.synchronize()
