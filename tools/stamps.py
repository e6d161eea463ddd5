"""Where does a short GEMM launch spend its time?  Per-block phase stamps of gemm_conv_kernel (developer build with
-DMF_STAMPS: tools/build_stamped.sh; run with MFHIP_LIB=reflecting-reality_amd/lib/libmfhip_stamps.so).

Slots (100 MHz real-time counter, wave 0 of each block; slots 8+ = the first staging wave of a warp-specialised block):
  0 kernel entry   1 prologue done (before the first DMA issue)   2 first K tile landed (barrier #0, warp-specialised tiles)
  3 main loop done   4 block barrier passed   5 epilogue stores have left   6 exit
Reported relative to the EARLIEST block's entry: when do blocks start (ramp), and how long is each phase (median / max)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops  # noqa: E402

hip.AUTOTUNE = False
prec = ops.Precision.get(os.environ.get("MF_STAMPS_PREC", "bf16"))      # MF_STAMPS_PREC=f16x3: the parity mode's kernels
lib = hip.load()
if not hasattr(lib, "mf_debug_set_stamps"):
    sys.exit("run with MFHIP_LIB pointing at the stamped build (tools/build_stamped.sh)")

CASES = [  # (label, batch, h, w, cin, cout, k, tile, splitk, with residual[, with bias, with temb])
    ("to_out 64^2", 8, 64, 64, 320, 320, 1, 26, 1, True), ("to_out 64^2 t2", 8, 64, 64, 320, 320, 1, 2, 1, True),
    ("to_out 64^2 t48", 8, 64, 64, 320, 320, 1, 48, 1, True),
    ("proj 32^2", 8, 32, 32, 640, 640, 1, 48, 1, True), ("proj 16^2", 8, 16, 16, 1280, 1280, 1, 44, 1, True),
    ("proj 16^2 t3", 8, 16, 16, 1280, 1280, 1, 3, 1, True), ("proj 8^2", 8, 8, 8, 1280, 1280, 1, 3, 1, True),
    ("conv 64^2", 8, 64, 64, 320, 320, 3, 39, 1, True), ("conv 32^2", 8, 32, 32, 640, 640, 3, 48, 1, True),
    ("conv 16^2", 8, 16, 16, 1280, 1280, 3, 48, 2, True), ("conv 8^2", 8, 8, 8, 1280, 1280, 3, 44, 6, True),
    # 11-: what the epilogue's operands cost (bias / residual / temb on and off)
    ("proj 16^2 no bias", 8, 16, 16, 1280, 1280, 1, 44, 1, True, False), ("proj 16^2 no bias no res", 8, 16, 16, 1280, 1280, 1, 44, 1, False, False),
    ("proj 16^2 bias no res", 8, 16, 16, 1280, 1280, 1, 44, 1, False, True),
    ("proj 32^2 no bias no res", 8, 32, 32, 640, 640, 1, 48, 1, False, False),
    ("conv 64^2 no bias no res", 8, 64, 64, 320, 320, 3, 39, 1, False, False), ("conv 64^2 bias+temb", 8, 64, 64, 320, 320, 3, 39, 1, False, True, True),
    ("conv 64^2 bias+temb+res", 8, 64, 64, 320, 320, 3, 39, 1, True, True, True),
    ("to_out 64^2 no bias no res", 8, 64, 64, 320, 320, 1, 26, 1, False, False),
    # 19-: round 5, the cross-tile fragment pipeline and the 64x80 wave tiles
    ("conv 64^2 t49", 8, 64, 64, 320, 320, 3, 49, 1, True), ("conv 64^2 960 t49", 8, 64, 64, 960, 320, 3, 49, 1, True),
    ("conv 32^2 t48", 8, 32, 32, 640, 640, 3, 48, 1, True), ("conv 32^2 t52", 8, 32, 32, 640, 640, 3, 52, 1, True),
    ("conv 32^2 t47", 8, 32, 32, 640, 640, 3, 47, 1, True), ("conv 32^2 1280 t48", 8, 32, 32, 1280, 640, 3, 48, 1, True),
    ("conv 16^2 t48 sk2", 8, 16, 16, 1280, 1280, 3, 48, 2, True), ("proj 32^2 t48", 8, 32, 32, 640, 640, 1, 48, 1, True),
    ("ff-out 64^2 1280->320 t48", 8, 64, 64, 1280, 320, 1, 48, 1, True),
    # 28-: the parity mode (MF_STAMPS_PREC=f16x3)
    ("conv 64^2 t37", 8, 64, 64, 320, 320, 3, 37, 1, True), ("conv 32^2 t14", 8, 32, 32, 640, 640, 3, 14, 1, True),
    ("conv 32^2 t38", 8, 32, 32, 640, 640, 3, 38, 1, True), ("conv 32^2 t41", 8, 32, 32, 640, 640, 3, 41, 1, True),
    ("proj 32^2 t41", 8, 32, 32, 640, 640, 1, 41, 1, True),
    # 33-: the round-5 loop forms as separate tiles (53-66) against their round-4 loops, and the short-K feed-forward projection
    ("conv 64^2 t59", 8, 64, 64, 320, 320, 3, 59, 1, True), ("conv 64^2 t63", 8, 64, 64, 320, 320, 3, 63, 1, True),
    ("conv 32^2 t54", 8, 32, 32, 640, 640, 3, 54, 1, True), ("conv 32^2 t53", 8, 32, 32, 640, 640, 3, 53, 1, True),
    ("ff-in 64^2 320->2560 t29", 8, 64, 64, 320, 2560, 1, 29, 1, False), ("ff-in 64^2 320->2560 t48", 8, 64, 64, 320, 2560, 1, 48, 1, False),
    ("ff-in 64^2 320->2560 t14", 8, 64, 64, 320, 2560, 1, 14, 1, False),
]


def run(label, b, h, w, ci, co, k, tile, sk, with_res, with_bias=True, with_temb=False):
    x = torch.randn(b, h, w, ci, device="cuda").to(prec.act)
    cw = ops.ConvWeight(torch.randn(co, ci, k, k) * 0.02, torch.randn(co) if with_bias else None, prec, "cuda")
    res = torch.randn(b, h, w, co, device="cuda").to(prec.act) if with_res else None
    temb = torch.randn(b, co, device="cuda") if with_temb else None
    other = torch.randn(b, h, w, co, device="cuda").to(prec.act)
    fn = lambda: ops.conv2d(x, cw, padding=k // 2, tile=tile, splitk=sk, res0=res, temb=temb)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    nblk = 8192
    buf = torch.zeros(nblk, 32, dtype=torch.int64, device="cuda")
    lib.mf_debug_set_stamps(C.c_void_p(buf.data_ptr()))
    # the launch under test sits between two other kernels of a replayed graph, like in the denoise step
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        other.add_(1.0)
        fn()
        other.add_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    buf.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    lib.mf_debug_set_stamps(C.c_void_p(0))
    st = buf.cpu().numpy().astype(np.int64)
    used = st[:, 0] > 0
    st = st[used]
    t0 = st[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    n = st.shape[0]
    ws = (st[:, 16] > 0).any()
    print(f"=== {label}: M={b * h * w} N={co} K={ci * k * k} tile {tile} split-K {sk}: {n} blocks", flush=True)
    ent = us(st[:, 0])
    print(f"   block entry: median {np.median(ent):6.2f} us, 90 % {np.percentile(ent, 90):6.2f}, last {ent.max():6.2f}")
    names = ["entry", "prologue", "tile 0 landed", "main loop", "barrier", "epilogue stores", "exit"]
    prev = st[:, 0]
    for s_ in range(1, 7):
        cur = st[:, s_]
        ok = cur > 0
        if not ok.any():
            continue
        d = (cur[ok] - prev[ok]) / 100.0
        print(f"   -> {names[s_]:16s}: median {np.median(d):6.2f} us  max {d.max():6.2f}   (reached at: median {np.median(us(cur[ok])):6.2f}, last {us(cur[ok]).max():6.2f})")
        prev = np.where(ok, cur, prev)
    if ws:
        p = st[:, 16:]
        ok = p[:, 0] > 0
        print(f"   staging wave: prologue {np.median((p[ok, 1] - p[ok, 0]) / 100.0):5.2f} us, tile 0 landed after {np.median((p[ok, 2] - p[ok, 1]) / 100.0):5.2f} us (max {((p[ok, 2] - p[ok, 1]) / 100.0).max():5.2f})")
    for ih in range(2):                         # warp-specialised epilogue rounds: slab written / barrier passed / stores issued
        a, b_, c = st[:, 7 + 3 * ih], st[:, 8 + 3 * ih], st[:, 9 + 3 * ih]
        if (a > 0).any():
            ref_t = st[:, 4] if ih == 0 else st[:, 9]
            print(f"   epilogue round {ih} (wave 0): slab written +{np.median((a - ref_t) / 100.0):5.2f} us, barrier +{np.median((b_ - a) / 100.0):5.2f}, "
                  f"chunks read / finished / stores issued +{np.median((c - b_) / 100.0):5.2f}")
            if ws:
                pa, pb, pc = st[:, 16 + 7 + 3 * ih], st[:, 16 + 8 + 3 * ih], st[:, 16 + 9 + 3 * ih]
                print(f"      staging wave 0: barrier reached at +{np.median((pb - (st[:, 4] if ih == 0 else st[:, 9])) / 100.0):5.2f} us of the round, stores issued +{np.median((pc - pb) / 100.0):5.2f}")
    if ws and (st[:, 14] > 0).any():             # main-loop wait shares (shader clocks): who waits for whom
        ok = st[:, 14] > 0
        loop, wbar, ntap = st[ok, 14].astype(float), st[ok, 13].astype(float), st[ok, 15].astype(float)
        p_vm, p_bar, p_iss = st[ok, 16 + 13].astype(float), st[ok, 16 + 14].astype(float), st[ok, 16 + 15].astype(float)
        print(f"   main loop: {np.median(loop / ntap):6.0f} clocks per K tile over {int(np.median(ntap))} tiles; compute wave 0 waits at the barrier "
              f"{np.median(wbar / loop) * 100:4.1f} % of it; staging wave 0: issuing {np.median(p_iss / loop) * 100:4.1f} %, waiting for its DMAs "
              f"{np.median(p_vm / loop) * 100:4.1f} %, at the barrier {np.median(p_bar / loop) * 100:4.1f} %")
    print(f"   launch span (first entry -> last exit): {us(st[:, 6]).max():6.2f} us", flush=True)


if __name__ == "__main__":
    sel = [int(i) for i in sys.argv[1].split(",")] if len(sys.argv) > 1 else range(len(CASES))
    for i in sel:
        run(*CASES[i])
