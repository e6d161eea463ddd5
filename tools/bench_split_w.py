"""f16x3 conv / GEMM at the training step's shapes: fp32 weights split in registers (w_split=0, what training runs) against
weights pre-split into (hi, lo) halves (w_split=1, what the inference parity mode runs).  Autotuned tile for each."""
import sys, os, importlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("reflecting-reality_amd.hip")
ops = importlib.import_module("reflecting-reality_amd.ops")
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
                              (8, 64, 320, 1280, 1), (8, 64, 1280, 320, 1), (8, 32, 640, 640, 1), (8, 32, 1920, 640, 3), (8, 8, 1280, 1280, 3)):
    x = torch.randn(b, hw, hw, cin, device=dev)
    w = torch.randn(cout, k * k * cin, device=dev) * 0.02
    wp, kp = ops.split_pack(w, hip.MF_F16X3)
    out = torch.empty(b, hw, hw, cout, device=dev)
    r = []
    for ws, ww, ld in ((0, w, k * k * cin), (1, wp, kp)):
        r.append(t(lambda: hip.gemm_conv(x, ww, out, dtype=hip.MF_F16X3, w_split=ws, ldw=ld, c0=cin, lda0=cin, batch=b, h_in=hw, w_in=hw,
                                         h_out=hw, w_out=hw, kh=k, kw=k, pad_t=k // 2, pad_l=k // 2, n=cout)))
    fl = 2.0 * b * hw * hw * cin * cout * k * k
    print(f"f16x3 b{b} {hw}x{hw} {cin}->{cout} k{k}: raw fp32 weights {r[0]:.1f} us ({fl / r[0] / 1e6:.0f} TF/s)  pre-split {r[1]:.1f} us ({fl / r[1] / 1e6:.0f} TF/s)")
