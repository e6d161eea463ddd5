"""Time mf_conv_wgrad (f16x3) at the training step's shapes (batch 8 x 512^2).  MFHIP_WGRAD_V1=1 selects the first version."""
import sys, os, importlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("reflecting-reality_amd.hip")
dev = "cuda:0"

def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

tag = "v1" if os.environ.get("MFHIP_WGRAD_V1") else "v2"
for (b, hw, cin, cout, k) in ((8, 64, 320, 320, 3), (8, 64, 320, 320, 1), (8, 32, 640, 640, 3), (8, 16, 1280, 1280, 3), (8, 64, 960, 320, 3),
                              (8, 32, 640, 640, 1), (8, 8, 1280, 1280, 3)):
    x = torch.randn(b, hw, hw, cin, device=dev); dy = torch.randn(b * hw * hw, cout, device=dev)
    dw = torch.zeros(cout, k * k * cin, device=dev)
    us = t(lambda: hip.conv_wgrad(x, dy, dw, code=hip.MF_F16X3, c0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw, w_out=hw, kh=k, kw=k,
                                  pad_t=k // 2, pad_l=k // 2, n=cout))
    fl = 2.0 * b * hw * hw * cin * cout * k * k
    print(f"wgrad[{tag}] b{b} {hw}x{hw} {cin}->{cout} k{k}: {us:.1f} us = {fl / us / 1e6:.0f} TF/s algorithmic (x3 MFMAs)")
