"""Epilogue cost of mf_gemm_conv: K=64 (one K-tile) GEMMs with / without residual and bias, graph-replayed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
for (m, n) in [(32768, 320), (8192, 640), (2048, 1280)]:
    for tile in (14, 2, 1):
        row = []
        for k in (64, 320):
            for (use_res, use_bias) in ((True, True), (False, True), (False, False)):
                x = torch.randn(m, k, device="cuda").bfloat16()
                w = ops.ConvWeight(torch.randn(n, k) * 0.05, torch.randn(n) if use_bias else None, prec, "cuda")
                res = torch.randn(m, n, device="cuda").bfloat16() if use_res else None
                out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
                row.append(timed(lambda: ops.linear(x, w, res0=res, tile=tile, splitk=1, out=out)))
        print(f"M={m} N={n} tile={tile}: K=64 res+bias/bias/none, K=320 same: " + " ".join(f"{t:6.1f}" for t in row), flush=True)
