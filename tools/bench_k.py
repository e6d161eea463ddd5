"""Fixed vs per-K cost of mf_gemm_conv: time(M, N, K) for a K sweep, per tile.  Launches are replayed from a
hipGraph so the host launch rate (~16 us per ctypes call) does not floor the numbers."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
REP = 20


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP


if __name__ == "__main__":
    tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1, 14, 2, 6, 3)
    for (m, n) in [(32768, 320), (2048, 1280), (8192, 640), (512, 1280)]:
        for tile in tiles:
            row = []
            for k in (64, 320, 640, 1280, 2560):
                x = torch.randn(m, k, device="cuda").bfloat16()
                w = ops.ConvWeight(torch.randn(n, k) * 0.05, torch.randn(n), prec, "cuda")
                res = torch.randn(m, n, device="cuda").bfloat16()
                out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
                row.append(timed(lambda: ops.linear(x, w, res0=res, tile=tile, splitk=1, out=out)))
            print(f"M={m} N={n} tile={tile}: us for K=64,320,640,1280,2560: " + " ".join(f"{t:7.1f}" for t in row), flush=True)
