"""Weight-gradient kernel (mf_conv_wgrad, bf16 inputs) against the forward GEMM of the same conv, per BrushNet shape at batch 8.
   python tools/bench_wgrad.py      (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reflecting_reality_amd import hip

DEV = "cuda"
SHAPES = [  # (name, batch, hw, cin, cout, k)
    ("64^2 3x3 320->320", 8, 64, 320, 320, 3), ("64^2 1x1 320->320", 8, 64, 320, 320, 1),
    ("32^2 3x3 320->640", 8, 32, 320, 640, 3), ("32^2 3x3 640->640", 8, 32, 640, 640, 3), ("32^2 1x1 640->640", 8, 32, 640, 640, 1),
    ("16^2 3x3 640->1280", 8, 16, 640, 1280, 3), ("16^2 3x3 1280->1280", 8, 16, 1280, 1280, 3), ("16^2 1x1 1280->1280", 8, 16, 1280, 1280, 1),
    ("8^2 3x3 1280->1280", 8, 8, 1280, 1280, 3), ("8^2 1x1 1280->1280", 8, 8, 1280, 1280, 1),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, b, hw, cin, cout, k in SHAPES:
    m = b * hw * hw
    x = torch.randn(b, hw, hw, cin, device=DEV).bfloat16()
    dy = torch.randn(m, cout, device=DEV).bfloat16()
    w = torch.randn(cout, k * k * cin, device=DEV).bfloat16()
    dw = torch.zeros(cout, k * k * cin, device=DEV)
    out = torch.empty(m, cout, dtype=torch.bfloat16, device=DEV)
    pad = k // 2
    fwd = lambda: hip.gemm_conv(x, w, out, dtype=hip.MF_BF16, c0=cin, lda0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw, w_out=hw, kh=k, kw=k,
                                pad_t=pad, pad_l=pad, n=cout)
    wg = lambda: hip.conv_wgrad(x, dy, dw, code=hip.MF_BF16, c0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw, w_out=hw, kh=k, kw=k, pad_t=pad,
                                pad_l=pad, n=cout)
    gf = 2.0 * m * cout * k * k * cin / 1e9
    tf, tw = timeit(fwd), timeit(wg)
    print(f"{name:24s} {gf:7.1f} GFLOP  forward {tf:7.1f} us ({gf / tf * 1e3:6.0f} TF/s)   wgrad {tw:7.1f} us ({gf / tw * 1e3:6.0f} TF/s)   ratio {tw / tf:4.2f}")
