"""Run the flash attention kernel at the step's dominant shape a few times (for rocprofv3 --pmc passes):
self-attention, batch 8 (B_eff), 8 heads x d = 40, 4096 tokens, bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

prec = ops.Precision.get("bf16")
b, s, heads, d = 8, 4096, 8, 40
c = heads * d
q = torch.randn(b, s, c, device="cuda").bfloat16()
k = torch.randn(b, s, c, device="cuda").bfloat16()
vt = torch.randn(b, c, s, device="cuda").bfloat16()
for _ in range(5):
    ops.attention(q, k, vt, heads, s, d ** -0.5, prec)
torch.cuda.synchronize()
