"""A/B of GroupNorm statistics from the producing conv's epilogue (round 6): the pairs (conv -> GroupNorm) of one denoise step at
batch 8, graph-replayed, with the statistics computed by the GroupNorm's own pass (MFHIP_GN_FROM_PARTS semantics off: gn_part=False)
and handed over by the conv (gn_part=32).  Prints conv alone, conv + stats, GroupNorm both ways and the pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402
from ab_ops import timed  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
prec = ops.Precision.get("bf16")


def main():
    torch.manual_seed(0)
    # (hw side, cin of the conv, cout = GroupNorm channels of segment 0, skip channels of segment 1, kernel)
    for (side, cin, cout, c1, ks) in ((64, 320, 320, 0, 3), (64, 320, 320, 320, 3), (64, 640, 320, 320, 3), (32, 640, 640, 0, 3),
                                      (32, 640, 640, 640, 3), (32, 1280, 640, 320, 3), (64, 320, 320, 0, 1), (32, 640, 640, 0, 1)):
        b = 8
        x = torch.randn(b, side, side, cin, device=dev).to(BF)
        res = torch.randn(b, side, side, cout, device=dev).to(BF)
        cw = ops.ConvWeight(torch.randn(cout, cin, ks, ks, device=dev) * 0.02, torch.zeros(cout, device=dev), prec, dev)
        skip = torch.randn(b, side, side, c1, device=dev).to(BF) if c1 else None
        if skip is not None:
            cw1 = ops.ConvWeight(torch.randn(c1, c1, 1, 1, device=dev) * 0.02, torch.zeros(c1, device=dev), prec, dev)
            sk_plain = ops.conv2d(skip, cw1, padding=0)
            sk_part = ops.conv2d(skip, cw1, padding=0, gn_part=32)
        g, be = torch.ones(cout + c1, device=dev), torch.zeros(cout + c1, device=dev)
        ops.conv2d(x, cw, padding=ks // 2, res0=res)                               # autotune outside the timed graphs
        y0 = ops.conv2d(x, cw, padding=ks // 2, res0=res)
        y1 = ops.conv2d(x, cw, padding=ks // 2, res0=res, gn_part=32)
        t_c0 = timed(lambda: ops.conv2d(x, cw, padding=ks // 2, res0=res))
        t_c1 = timed(lambda: ops.conv2d(x, cw, padding=ks // 2, res0=res, gn_part=32))
        t_g0 = timed(lambda: hip.groupnorm(y0, g, be, groups=32, eps=1e-5, silu=True, out_dtype=BF, x1=sk_plain if c1 else None))
        t_g1 = timed(lambda: hip.groupnorm(y1, g, be, groups=32, eps=1e-5, silu=True, out_dtype=BF, x1=sk_part if c1 else None))
        print(f"{side}x{side} conv{ks} {cin}->{cout} (+skip {c1}): conv {t_c0:6.1f} -> {t_c1:6.1f} us (rows/block {y1._gn_part[1]}), "
              f"groupnorm {t_g0:6.1f} -> {t_g1:6.1f} us, pair {t_c0 + t_g0:6.1f} -> {t_c1 + t_g1:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
