"""The feed-forward input projection of the step (LayerNorm folded in, GEGLU epilogue: attention.py:1188-1201, activations.py:85-103) and the
other short-K 1x1 GEMMs, per tile, graph-timed: the persistent tile 69 (csrc/gemm_nloop.hip) against the tiles the tuner picks today."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402
from bench_k import timed  # noqa: E402

hip.AUTOTUNE = False
dev = torch.device("cuda:0")
prec = ops.Precision.get(sys.argv[1] if len(sys.argv) > 1 else "bf16")
g = torch.Generator().manual_seed(0)
TILES = (70, 69, 29, 30, 14, 26, 48, 54, 67, 6, 3, 1)
print("precision", prec.name)
for rows, c in ((32768, 320), (8192, 640), (2048, 1280)):
    x = (torch.randn(rows, c, generator=g) * 1.3).to(dev, prec.act)
    gamma, beta = torch.ones(c), torch.zeros(c)
    w = torch.randn(8 * c, c, generator=g) / c ** 0.5
    gw0 = ops.geglu_weight(w, torch.zeros(8 * c), prec, dev)
    line = f"ff1 geglu (the step's form) {rows}x{c}->{8 * c}:"
    for t in TILES:
        try:
            us = timed(lambda: ops.linear_geglu(x, gw0, tile=t))
            line += f"  t{t} {us:6.1f}"
        except hip.MfhipError:
            pass
    print(line, flush=True)
    gw = ops.geglu_weight(w, torch.zeros(8 * c), prec, dev, ln=(gamma, beta, 1e-5))
    line = f"ff1 ln+geglu {rows}x{c}->{8 * c}:"
    for t in TILES:
        try:
            us = timed(lambda: ops.linear_geglu(x, gw, tile=t))
            line += f"  t{t} {us:6.1f}"
        except hip.MfhipError:
            pass
    print(line, flush=True)
    # the same GEMM with nothing for the epilogue to load (no bias, no LayerNorm fold): what the loads in the storing waves cost
    lw0 = ops.ConvWeight(w[:4 * c], None, prec, dev)
    line = f"plain {rows}x{c}->{4 * c}, no bias:"
    for t in TILES:
        try:
            us = timed(lambda: ops.linear(x, lw0, tile=t))
            line += f"  t{t} {us:6.1f}"
        except hip.MfhipError:
            pass
    print(line, flush=True)
    # the block's other Linears: to_q (folded LayerNorm), to_out / proj_out (+ residual), ff.net.2 (K = 4 C, + residual)
    res = torch.randn(rows, c, generator=g).to(dev, prec.act)
    for label, k, n, ln, r in (("to_q ln", c, c, True, False), ("to_out +res", c, c, False, True), ("ff2 +res", 4 * c, c, False, True)):
        if n % 160:
            continue
        xx = (torch.randn(rows, k, generator=g)).to(dev, prec.act)
        lw = ops.ConvWeight(torch.randn(n, k, generator=g) / k ** 0.5, torch.zeros(n), prec, dev, ln=(torch.ones(k), torch.zeros(k), 1e-5) if ln else None)
        line = f"{label} {rows}x{k}->{n}:"
        for t in TILES:
            try:
                us = timed(lambda: ops.linear(xx, lw, tile=t, res0=res if r else None))
                line += f"  t{t} {us:6.1f}"
            except hip.MfhipError:
                pass
        print(line, flush=True)
