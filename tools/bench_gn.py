"""GroupNorm shapes of one denoise step (batch 8 = CFG x 4 images), graph-replayed; run under rocprofv3 --kernel-trace
--stats for the per-kernel split."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip
from bench_k import timed

for (b_, hw, c0, c1) in ((8, 4096, 320, 0), (8, 4096, 320, 320), (8, 1024, 640, 0), (8, 1024, 640, 320), (8, 256, 1280, 0),
                         (8, 256, 1280, 1280), (8, 64, 1280, 0), (8, 64, 1280, 1280)):
    x = torch.randn(b_, hw, c0, device="cuda").bfloat16()
    x1 = torch.randn(b_, hw, c1, device="cuda").bfloat16() if c1 else None
    c = c0 + c1
    g = torch.ones(c, device="cuda"); be = torch.zeros(c, device="cuda")
    t = timed(lambda: hip.groupnorm(x, g, be, groups=32, eps=1e-5, silu=True, out_dtype=torch.bfloat16, x1=x1))
    print(f"groupnorm B={b_} HW={hw} C={c0}+{c1}: {t:7.1f} us ({6 * b_ * hw * c / t / 1e6:5.2f} TB/s for read+read+write)", flush=True)
