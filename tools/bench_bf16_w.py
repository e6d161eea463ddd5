"""Training GEMMs of the bf16x1 mode (fp32 storage, bf16 products): MF_BF16X1 with raw fp32 weights (rounded in registers) against
MF_BF16 compute on fp32 activations with a bf16 copy of the weight (the A_F32 kernel forms)."""
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

for (b, hw, cin, cout, k) in ((8, 64, 320, 320, 3), (8, 64, 320, 320, 1), (8, 32, 640, 640, 3), (8, 16, 1280, 1280, 3), (8, 64, 960, 320, 3),
                              (8, 64, 320, 1280, 1), (8, 64, 1280, 320, 1), (8, 32, 640, 640, 1), (8, 8, 1280, 1280, 3)):
    x = torch.randn(b, hw, hw, cin, device=dev)
    w = torch.randn(cout, k * k * cin, device=dev) * 0.02
    wb = w.to(torch.bfloat16)
    o1 = torch.empty(b, hw, hw, cout, device=dev); o2 = torch.empty_like(o1)
    kw = dict(c0=cin, lda0=cin, batch=b, h_in=hw, w_in=hw, h_out=hw, w_out=hw, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, n=cout)
    r1 = t(lambda: hip.gemm_conv(x, w, o1, dtype=hip.MF_BF16X1, **kw))
    r2 = t(lambda: hip.gemm_conv(x, wb, o2, dtype=hip.MF_BF16, **kw))
    fl = 2.0 * b * hw * hw * cin * cout * k * k
    err = float((o1 - o2).abs().max() / o1.abs().max())
    print(f"b{b} {hw}x{hw} {cin}->{cout} k{k}: bf16x1 (fp32 W) {r1:.1f} us ({fl / r1 / 1e6:.0f} TF/s)  bf16 W copy {r2:.1f} us ({fl / r2 / 1e6:.0f} TF/s)  max rel diff {err:.1e}")
