"""Same-box A/B of single launches between source trees: run from the root of a tree (`cd _r04 && python ../tools/ab_ops.py`);
prints one line per case, graph-timed (N launches of one op over a ring of buffers larger than L2 + Infinity Cache is NOT
attempted: the step's own operands are cache-warm too, so these are warm numbers)."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from reflecting_reality_amd import hip, ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timed(fn, n=40, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
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


def main():
    torch.manual_seed(0)
    rows = []
    for (b, hw, c0, c1) in ((2, 4096, 320, 0), (2, 4096, 320, 320), (2, 1024, 640, 640), (2, 256, 1280, 1280), (2, 64, 1280, 1280), (2, 1024, 640, 0)):
        x0 = torch.randn(b, hw, c0, device=dev).to(BF)
        x1 = torch.randn(b, hw, c1, device=dev).to(BF) if c1 else None
        g = torch.ones(c0 + c1, device=dev)
        be = torch.zeros(c0 + c1, device=dev)
        out = torch.empty(b, hw, c0 + c1, device=dev, dtype=BF)
        rows.append((f"groupnorm bf16 b{b} hw{hw} c{c0}+{c1} silu", timed(lambda: hip.groupnorm(x0, g, be, groups=32, eps=1e-5, silu=True, out_dtype=BF, x1=x1, out=out))))
    for (r, c) in ((8192, 320), (2048, 640), (512, 1280)):
        x = torch.randn(r, c, device=dev).to(BF)
        g = torch.ones(c, device=dev)
        be = torch.zeros(c, device=dev)
        rows.append((f"layernorm bf16 {r}x{c}", timed(lambda: hip.layernorm(x, g, be, 1e-5, BF))))
    # convs / linears at fixed tiles (no autotune): 3x3 conv 64^2 320->320, 32^2 640->640, linear 8192x320->1280(x2 geglu-ish plain)
    for (hw, cin, cout, ks, tiles) in ((64, 320, 320, 3, (26, 48, 20, 42)), (32, 640, 640, 3, (48, 40, 27)), (16, 1280, 1280, 3, (48, 27)),
                                       (64, 320, 320, 1, (26, 29)), (32, 640, 640, 1, (48, 29))):
        x = torch.randn(2, hw, hw, cin, device=dev).to(BF)
        w = (torch.randn(cout, cin, ks, ks, device=dev) * 0.02)
        prec = ops.Precision.get("bf16")
        try:
            cw = ops.ConvWeight(w, torch.zeros(cout, device=dev), prec, dev)
        except Exception as e:  # constructor signature differs between trees: report and go on
            rows.append((f"conv {ks}x{ks} {hw}^2 {cin}->{cout}: ConvWeight {e!r}", float("nan")))
            continue
        for t in tiles:
            try:
                rows.append((f"conv {ks}x{ks} {hw}^2 {cin}->{cout} tile {t}", timed(lambda: ops.conv2d(x, cw, tile=t, padding=ks // 2))))
            except Exception as e:
                rows.append((f"conv {ks}x{ks} {hw}^2 {cin}->{cout} tile {t}: {str(e)[:60]}", float("nan")))
    for name, us in rows:
        print(f"{us:9.2f} us  {name}")


if __name__ == "__main__":
    main()
