"""Calibrate the per-launch floor inside a hipGraph: tiny kernel, streaming add, torch copy, GroupNorm."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip
from bench_k import timed

for n in (64, 1 << 20, 32768 * 320, 32768 * 1280, 1 << 28):
    a = torch.randn(n, device="cuda").bfloat16(); b = torch.randn(n, device="cuda").bfloat16()
    o = torch.empty_like(a)
    t = timed(lambda: hip.add(a, b, torch.bfloat16))
    t2 = timed(lambda: o.copy_(a))
    t3 = timed(lambda: torch.add(a, b, out=o))
    print(f"n={n:>10d} bf16: mf_add {t:7.1f} us ({6 * n / t / 1e6:6.2f} TB/s)  torch.copy_ {t2:7.1f} us ({4 * n / t2 / 1e6:6.2f} TB/s)"
          f"  torch.add {t3:7.1f} us ({6 * n / t3 / 1e6:6.2f} TB/s)", flush=True)
for (b_, hw, c) in ((8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280)):
    x = torch.randn(b_, hw, c, device="cuda").bfloat16()
    g = torch.ones(c, device="cuda"); be = torch.zeros(c, device="cuda")
    t = timed(lambda: hip.groupnorm(x, g, be, groups=32, eps=1e-5, silu=True, out_dtype=torch.bfloat16))
    print(f"groupnorm B={b_} HW={hw} C={c}: {t:7.1f} us ({6 * x.numel() / t / 1e6:5.2f} TB/s for read+read+write)")
for (rows, c) in ((32768, 320), (8192, 640), (2048, 1280)):
    x = torch.randn(rows, c, device="cuda").bfloat16()
    g = torch.ones(c, device="cuda"); be = torch.zeros(c, device="cuda")
    t = timed(lambda: hip.layernorm(x, g, be, 1e-5, torch.bfloat16))
    print(f"layernorm rows={rows} C={c}: {t:7.1f} us ({4 * rows * c / t / 1e6:5.2f} TB/s for read+write)")
