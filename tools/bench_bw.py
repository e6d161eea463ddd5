"""HBM GB/s of the bandwidth-bound helper kernels of the denoise step at their production shapes (batch 4 x 512x512, CFG:
B_eff = 8), HIP events over graph-free back-to-back launches.  Algorithmic bytes = every input read once + every output
written once (GroupNorm: the input twice — statistics, then apply).  Prints a markdown table (profiles/r03_hbm_bandwidth.md).

    python3 tools/bench_bw.py > gpurun_out/r03_hbm_bandwidth.md
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
PEAK = 8000.0


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e-3


rows = []


def add(name, shape, nbytes, fn):
    t = timeit(fn)
    rows.append((name, shape, nbytes / 1e6, t * 1e6, nbytes / t / 1e9))


hip.load()
for b, hw, c in ((8, 64 * 64, 320), (8, 64 * 64, 640), (8, 32 * 32, 640), (8, 32 * 32, 1280), (8, 16 * 16, 1280), (8, 16 * 16, 2560), (8, 8 * 8, 1280)):
    x = torch.randn(b, hw, c, device=DEV).to(BF)
    g, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    add("GroupNorm+SiLU (gn_stats + gn_apply)", f"{b}x{hw}x{c} bf16", 3 * x.numel() * 2,
        lambda: hip.groupnorm(x, g, be, groups=32, eps=1e-5, silu=True, out_dtype=BF))
for rows_, c in ((8 * 4096, 320), (8 * 1024, 640), (8 * 256, 1280)):
    x = torch.randn(rows_, c, device=DEV).to(BF)
    g, be = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    add("LayerNorm (layernorm8)", f"{rows_}x{c} bf16", 2 * x.numel() * 2, lambda: hip.layernorm(x, g, be, 1e-5, BF))
    if c <= 2048:
        add("LayerNorm + fp8 quantise", f"{rows_}x{c} bf16 -> e4m3", x.numel() * 3, lambda: hip.quantize_rows_fp8(x, (g, be), 1e-5))
lat = torch.randn(4, 4, 64, 64, device=DEV)
eps = torch.randn(8, 4, 64, 64, device=DEV)
coef = torch.tensor([0.5, 0.8, 0.6, 0.7], device=DEV)
add("CFG + DDIM update (cfg_ddim_step_dev)", "4x4x64x64 fp32", (2 + 1 + 1) * lat.numel() * 4,
    lambda: hip.cfg_ddim_step_dev(eps[:4], eps[4:], 7.5, lat, coef, out=lat))
x = torch.randn(8, 320, 64, 64, device=DEV)
add("pack NCHW fp32 -> NHWC bf16 (pack_nhwc)", "8x320x64x64", x.numel() * 6, lambda: hip.pack_nhwc(x, None, 320, BF))
a, b2 = torch.randn(8, 64, 64, 320, device=DEV).to(BF), torch.randn(8, 64, 64, 320, device=DEV).to(BF)
add("residual add (mf_add bf16)", "8x64x64x320", 3 * a.numel() * 2, lambda: hip.add(a, b2, BF))
# split-K reduce: a deep-K low-resolution conv with a forced split
prec = ops.Precision.get("bf16")
xs = torch.randn(8, 8, 8, 2560, device=DEV).to(BF)
cw = ops.ConvWeight(torch.randn(1280, 2560, 3, 3) * 0.01, torch.zeros(1280), prec, DEV)
t1 = timeit(lambda: ops.conv2d(xs, cw, splitk=1, tile=1))
t8 = timeit(lambda: ops.conv2d(xs, cw, splitk=8, tile=1))
rows.append(("conv3x3 2560->1280 @8x8 split-K 8 incl. splitk_reduce (vs split-K 1)", "M=512 N=1280 K=23040", 8 * 512 * 1280 * 4 * 2 / 1e6, t8 * 1e6,
             float("nan")))
rows.append(("  same conv, split-K 1", "", 0.0, t1 * 1e6, float("nan")))
img = torch.rand(4, 3, 512, 512, device=DEV)
add("front-end: preprocess (min reduction + normalise)", "4x3x512x512 fp32", img.numel() * 4 * 3, lambda: hip.image_normalize(img))
add("front-end: postprocess to uint8 NHWC", "4x3x512x512", img.numel() * 5, lambda: hip.postprocess(img, uint8=True))
print("| kernel | shape | algorithmic MB | us per call | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|")
for name, shape, mb, us, gbs in rows:
    print(f"| {name} | {shape} | {mb:.1f} | {us:.1f} | {'' if gbs != gbs else f'{gbs:.0f}'} | {'' if gbs != gbs else f'{gbs / PEAK:.2f}'} |")
