"""Does splitting a launch over the batch into two concurrent half-batch launches (two streams of one hipGraph) beat the one launch?
One whole-CU block per CU runs prologue / loop / epilogue in lockstep over the chip; two blocks of different launches per CU overlap
one's epilogue with the other's loop.  Per shape: best tile of the full launch vs best tile of the concurrent halves."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

hip.AUTOTUNE = False
dev = torch.device("cuda:0")
BF = torch.bfloat16
prec = ops.Precision.get("bf16")
side = torch.cuda.Stream()


def timed(fn, n=10, reps=4):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000.0 / n)
    return best


CASES = [("conv 64^2 320->320", 64, 320, 320, 3), ("conv 64^2 640->320", 64, 640, 320, 3), ("conv 64^2 960->320", 64, 960, 320, 3),
         ("conv 32^2 640->640", 32, 640, 640, 3), ("conv 32^2 1280->640", 32, 1280, 640, 3), ("conv 32^2 1920->640", 32, 1920, 640, 3),
         ("conv 16^2 2560->1280", 16, 2560, 1280, 3), ("1x1 64^2 320->2560", 64, 320, 2560, 1), ("1x1 64^2 1280->320", 64, 1280, 320, 1),
         ("1x1 32^2 640->5120", 32, 640, 5120, 1), ("1x1 32^2 2560->640", 32, 2560, 640, 1)]
TILES = {3: (20, 21, 27, 28, 37, 39, 40, 47, 48, 49, 51, 53, 54, 55, 63, 14, 26), 1: (1, 14, 25, 26, 29, 30, 41, 43, 44, 48, 50, 52, 54, 56, 57, 64)}
for label, hw, cin, cout, ks in CASES:
    x = torch.randn(8, hw, hw, cin, device=dev).to(BF)
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.02
    cw = ops.ConvWeight(w, torch.zeros(cout, device=dev), prec, dev)
    xa, xb = x[:4], x[4:]
    full, half = {}, {}
    for t in TILES[ks]:
        for sk in ((1,) if hw >= 32 else (1, 2, 4)):
            try:
                full[(t, sk)] = timed(lambda: ops.conv2d(x, cw, tile=t, splitk=sk, padding=ks // 2))
            except hip.MfhipError:
                continue

            def two():
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.conv2d(xb, cw, tile=t, splitk=sk, padding=ks // 2)
                ops.conv2d(xa, cw, tile=t, splitk=sk, padding=ks // 2)
                cur.wait_stream(side)
            half[(t, sk)] = timed(two)
    bf = min(full, key=full.get)
    bh = min(half, key=half.get)
    print(f"{label:24s} one launch: {full[bf]:7.1f} us (tile {bf[0]}/sk{bf[1]})   two concurrent halves: {half[bh]:7.1f} us (tile {bh[0]}/sk{bh[1]})   "
          f"ratio {half[bh] / full[bf]:.3f}   same tile as full-best: {half[bf]:7.1f}", flush=True)
