"""Run the split-precision flash attention forward + backward at the training step's dominant shape a few times (for rocprofv3
--pmc passes): self-attention, batch 8, 8 heads x d = 40, 4096 tokens, through the tape like a transformer block does."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import autograd, hip, ops  # noqa: E402

prec = ops.Precision.get(sys.argv[1] if len(sys.argv) > 1 else "f16x3")      # "bf16x1": the single-plane bf16 kernels
b, s, heads, d = 8, 4096, 8, 40
c = heads * d
q, k, v = (torch.randn(b, s, c, device="cuda") for _ in range(3))
g = torch.randn(b, s, c, device="cuda")
for _ in range(4):
    tape = autograd.Tape(prec.tape_code)
    ops.TAPE = tape
    try:
        out = ops.attention_train(q, k, v, heads, d ** -0.5, prec)
    finally:
        ops.TAPE = None
    tape.add(out, g)
    tape.backward()
torch.cuda.synchronize()
