"""Fixed cost of the mf_gemm_conv epilogue: K = 64 launches with / without residual, bf16 / fp32 output."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
for (m, n, tile) in [(32768, 320, 14), (8192, 640, 14), (2048, 1280, 1), (32768, 320, 1)]:
    for k in (64, 320):
        x = torch.randn(m, k, device="cuda").bfloat16()
        w = ops.ConvWeight(torch.randn(n, k) * 0.05, torch.randn(n), prec, "cuda")
        res = torch.randn(m, n, device="cuda").bfloat16()
        res2 = torch.randn(m, n, device="cuda").bfloat16()
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        t0 = timed(lambda: ops.linear(x, w, tile=tile, splitk=1, out=out))
        t1 = timed(lambda: ops.linear(x, w, res0=res, tile=tile, splitk=1, out=out))
        t2 = timed(lambda: ops.linear(x, w, res0=res, res1=res2, tile=tile, splitk=1, out=out))
        print(f"M={m} N={n} K={k} tile={tile}: no residual {t0:.1f} us | res0 {t1:.1f} us | res0+res1 {t2:.1f} us", flush=True)
