"""Run mf_conv_wgrad at one of the training step's shapes a few times (for rocprofv3 --pmc passes):
   python tools/one_wgrad.py <batch> <hw> <cin> <cout> <k> [bf16]      (default f16x3 on fp32 tensors; bf16 = the bf16x1 mode's
   pre-rounded bf16 operands: the LDS-DMA kernel, or the 160-wide register-staged one where the cost model picks it)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip  # noqa: E402

b, hw, cin, cout, k = (int(v) for v in sys.argv[1:6])
b16 = len(sys.argv) > 6 and sys.argv[6] == "bf16"
x = torch.randn(b, hw, hw, cin, device="cuda")
dy = torch.randn(b * hw * hw, cout, device="cuda")
if b16:
    x, dy = x.bfloat16(), dy.bfloat16()
dw = torch.zeros(cout, k * k * cin, device="cuda")
for _ in range(5):
    hip.conv_wgrad(x, dy, dw, code=hip.MF_BF16 if b16 else hip.MF_F16X3, c0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw, w_out=hw, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, n=cout)
torch.cuda.synchronize()
