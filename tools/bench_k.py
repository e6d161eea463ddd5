"""Fixed vs per-K cost of mf_gemm_conv: time(M, N, K) for a K sweep, per tile."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
for (m, n) in [(32768, 320), (2048, 1280), (8192, 640)]:
    for tile in (1, 14, 2, 6):
        row = []
        for k in (64, 320, 640, 1280, 2560):
            x = torch.randn(m, k, device="cuda").bfloat16()
            w = ops.ConvWeight(torch.randn(n, k) * 0.05, torch.randn(n), prec, "cuda")
            res = torch.randn(m, n, device="cuda").bfloat16()
            for _ in range(3): ops.linear(x, w, res0=res, tile=tile, splitk=1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.linear(x, w, res0=res, tile=tile, splitk=1)
            e1.record(); e1.synchronize()
            row.append(e0.elapsed_time(e1) * 50)
        print(f"M={m} N={n} tile={tile}: us for K=64,320,640,1280,2560: " + " ".join(f"{t:7.1f}" for t in row))
