"""Time the GroupNorm / LayerNorm backward kernels at the training step's shapes (batch 8 x 512^2)."""
import sys, os, importlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("reflecting-reality_amd.hip")
dev = "cuda:0"

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (b, c0, c1, hw) in ((8, 320, 0, 4096), (8, 640, 0, 1024), (8, 1280, 0, 256), (8, 640, 320, 4096), (8, 1280, 1280, 256), (8, 1280, 0, 64)):
    x0 = torch.randn(b, hw, c0, device=dev); x1 = torch.randn(b, hw, c1, device=dev) if c1 else None
    dy = torch.randn(b, hw, c0 + c1, device=dev); g = torch.randn(c0 + c1, device=dev); be = torch.randn(c0 + c1, device=dev)
    r = [t(lambda s=s: hip.groupnorm_bwd(x0, dy, g, be, groups=32, eps=1e-5, silu=True, x1=x1, streaming=s)) for s in (False, True)]
    mb = (x0.numel() + (x1.numel() if c1 else 0)) * 4 * 3 / 1e6
    print(f"groupnorm_bwd b{b} c{c0}+{c1} hw{hw}: blocks {r[0]:.1f} us  streaming {r[1]:.1f} us  ({mb:.0f} MB min traffic = {mb / 8e3 * 1e3:.1f} us at 8 TB/s)")
for rows, c in ((32768, 320), (8192, 640), (2048, 1280)):
    x = torch.randn(rows, c, device=dev); dy = torch.randn(rows, c, device=dev); g = torch.randn(c, device=dev)
    us = t(lambda: hip.layernorm_bwd(x, dy, g, 1e-5))
    print(f"layernorm_bwd {rows}x{c}: {us:.1f} us ({rows * c * 12 / 1e6:.0f} MB min traffic)")
