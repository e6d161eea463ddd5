"""Run mf_conv_wgrad (f16x3) at one of the training step's shapes a few times (for rocprofv3 --pmc passes):
   python tools/one_wgrad.py <batch> <hw> <cin> <cout> <k>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip  # noqa: E402

b, hw, cin, cout, k = (int(v) for v in sys.argv[1:6])
x = torch.randn(b, hw, hw, cin, device="cuda")
dy = torch.randn(b * hw * hw, cout, device="cuda")
dw = torch.zeros(cout, k * k * cin, device="cuda")
for _ in range(5):
    hip.conv_wgrad(x, dy, dw, code=hip.MF_F16X3, c0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw, w_out=hw, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, n=cout)
torch.cuda.synchronize()
