"""Epilogue cost of mf_gemm_conv with residuals (graph replay).  MFHIP_NO_RES_PRE=1 in the environment switches the
residual prefetch (loads issued ahead of the LDS transposition) off for an A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
tag = "no-prefetch" if os.environ.get("MFHIP_NO_RES_PRE") else "prefetch"
for (b, h, w, ci, co, k, t) in [(8, 64, 64, 320, 320, 3, 20), (8, 64, 64, 320, 320, 3, 14), (8, 64, 64, 320, 320, 1, 14), (8, 64, 64, 64, 320, 1, 14),
                                (8, 32, 32, 640, 640, 3, 20), (8, 32, 32, 640, 640, 1, 6), (8, 16, 16, 1280, 1280, 1, 12)]:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    wt = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
    r = torch.randn(b, h, w, co, device="cuda").bfloat16()
    r2 = torch.randn(b, h, w, co, device="cuda").bfloat16()
    t0 = timed(lambda: ops.conv2d(x, wt, padding=k // 2, tile=t, splitk=1))
    t1 = timed(lambda: ops.conv2d(x, wt, padding=k // 2, tile=t, splitk=1, res0=r))
    t2 = timed(lambda: ops.conv2d(x, wt, padding=k // 2, tile=t, splitk=1, res0=r, res1=r2))
    print(f"{tag} M={b*h*w} N={co} K={ci*k*k} tile {t}: no residual {t0:.1f} us | res0 {t1:.1f} us | res0+res1 {t2:.1f} us", flush=True)
