"""In-launch split-K combine (mf_gemm_desc.sk_tickets) against the reduce launch on the denoise step's small-M shapes:
tile x split-K x {reduce launch, in-launch}, graph-replayed (20 launches per replay, same box, same process)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops
from bench_k import timed
hip.AUTOTUNE = False
prec = ops.Precision.get("bf16")
hip.sk_tickets("cuda")
SHAPES = [  # (batch, h, w, cin, cout, k)  -- B_eff = 8
    (8, 8, 8, 1280, 1280, 3), (8, 8, 8, 2560, 1280, 3), (8, 8, 8, 1280, 1280, 1), (8, 8, 8, 2560, 1280, 1),
    (8, 16, 16, 1280, 1280, 1), (8, 16, 16, 2560, 1280, 1), (8, 16, 16, 5120, 1280, 1), (8, 16, 16, 1280, 1280, 3),
    (8, 16, 16, 2560, 1280, 3), (8, 32, 32, 640, 640, 1), (8, 32, 32, 2560, 640, 1),
]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[1].split(",")]
for (b, h, w, ci, co, k) in SHAPES:
    x = torch.randn(b, h, w, ci, device="cuda").bfloat16()
    cw = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co), prec, "cuda")
    res = torch.randn(b, h, w, co, device="cuda").bfloat16()
    fl = 2.0 * b * h * w * ci * co * k * k
    print(f"--- M={b * h * w} N={co} K={ci * k * k} ({k}x{k})   1000 TF/s = {fl / 1e9:.1f} us", flush=True)
    best = {}
    for tile in (41, 43, 44, 48, 1, 2, 6, 3, 46):
        for fused in (False, True):
            if fused and tile == 46:
                continue
            row = []
            for sk in (1, 2, 3, 4, 6, 8, 12, 16):
                if fused and sk == 1:
                    row.append("        ")
                    continue
                try:
                    t = timed(lambda: ops.conv2d(x, cw, padding=k // 2, tile=tile, splitk=sk, res0=res, sk_fused=fused))
                    row.append(f"{sk:2d}:{t:5.1f}")
                    key = "fused" if fused else "reduce"
                    if t < best.get(key, (1e9,))[0]:
                        best[key] = (t, tile, sk)
                except hip.MfhipError:
                    row.append(f"{sk:2d}:  n/a")
            print(f"tile {tile:2d} {'in-launch' if fused else 'reduce   '}: " + " ".join(row), flush=True)
    print(f"    best reduce-launch {best.get('reduce')}   best in-launch {best.get('fused')}", flush=True)
