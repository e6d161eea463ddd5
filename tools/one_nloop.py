"""One shape on one tile for counter passes (tools/pmc_nloop.sh): plain 32768 x 320 -> 1280 GEMM, no bias."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

hip.AUTOTUNE = False
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 69
prec = ops.Precision.get("bf16")
x = torch.randn(32768, 320, device="cuda").bfloat16()
lw = ops.ConvWeight(torch.randn(1280, 320) / 18.0, None, prec, "cuda")
for _ in range(5):
    y = ops.linear(x, lw, tile=tile)
torch.cuda.synchronize()
