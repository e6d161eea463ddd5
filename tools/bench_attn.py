"""Time the fused attention kernel on the SD1.5 shapes (B_eff=8, 8 heads)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
prec = ops.Precision.get("bf16")
for (s, skv, d) in [(4096, 4096, 40), (4096, 77, 40), (1024, 1024, 80), (256, 256, 160)]:
    c = 8 * d
    q = torch.randn(8, s, c, device="cuda").bfloat16(); k = torch.randn(8, skv, c, device="cuda").bfloat16()
    vt = torch.zeros(8, c, (skv + 7) // 8 * 8, device="cuda").bfloat16(); vt[:, :, :skv] = torch.randn(8, c, skv, device="cuda").bfloat16()
    for _ in range(3): ops.attention(q, k, vt, 8, skv, 1 / math.sqrt(d), prec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attention(q, k, vt, 8, skv, 1 / math.sqrt(d), prec)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"S={s} Skv={skv} d={d}: {us:.1f} us  {4.0 * 8 * 8 * s * skv * d / us / 1e6:.0f} TF/s")
