"""GEGLU GEMMs of the three transformer levels (batch 8), graph-replayed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
prec = ops.Precision.get("bf16")
for (m, c) in ((32768, 320), (8192, 640), (2048, 1280)):
    x = torch.randn(m, c, device="cuda").bfloat16()
    lw = ops.geglu_weight(torch.randn(8 * c, c) * 0.05, torch.randn(8 * c), prec, "cuda")
    t = timed(lambda: ops.linear_geglu(x, lw))
    print(f"GEGLU M={m} N={8 * c} K={c}: {t:7.1f} us {2.0 * m * 8 * c * c / t / 1e6:5.0f} TF/s", flush=True)
