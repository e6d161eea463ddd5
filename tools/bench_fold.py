"""Folded LayerNorm / fused q|k|v GEMMs against the launches they replace, per level of the SD1.5 UNet (graph-replayed)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reflecting_reality_amd import hip, ops

prec = ops.Precision.get("bf16")
dev = "cuda"


def bench(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


TILES = (41, 42, 43, 44, 45, 46, 48)
for (b, s, c) in [(8, 4096, 320), (8, 1024, 640), (8, 256, 1280), (8, 64, 1280)]:
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(b, s, c, generator=g)).to(dev).bfloat16()
    gamma, beta = torch.ones(c), torch.zeros(c)
    nrm = (gamma.to(dev), beta.to(dev))
    wq, wk, wv = (torch.randn(c, c, generator=g) / c ** 0.5 for _ in range(3))
    w_qk = ops.ConvWeight(torch.cat([wq, wk], 0), None, prec, dev)
    w_v = ops.ConvWeight(wv, None, prec, dev, raw=True)
    w_qkv = ops.ConvWeight(torch.cat([wq, wk, wv], 0), None, prec, dev, ln=(gamma, beta, 1e-5))
    w_qkv_noln = ops.ConvWeight(torch.cat([wq, wk, wv], 0), None, prec, dev)
    w_q = ops.ConvWeight(wq, None, prec, dev)
    w_q_ln = ops.ConvWeight(wq, None, prec, dev, ln=(gamma, beta, 1e-5))
    wg, bg = torch.randn(8 * c, c, generator=g) / c ** 0.5, torch.randn(8 * c, generator=g)
    w_g = ops.geglu_weight(wg, bg, prec, dev)
    w_g_ln = ops.geglu_weight(wg, bg, prec, dev, ln=(gamma, beta, 1e-5))
    vt_buf = torch.empty(b, c, s, dtype=torch.bfloat16, device=dev)

    t_ln = bench(lambda: ops.layernorm(x, nrm, 1e-5, torch.bfloat16))
    n = ops.layernorm(x, nrm, 1e-5, torch.bfloat16)
    t_qk = bench(lambda: ops.linear(n, w_qk))
    t_vt = bench(lambda: ops.linear_t(n, w_v, s, out=vt_buf))
    t_q = bench(lambda: ops.linear(n, w_q))
    t_g = bench(lambda: ops.linear_geglu(n, w_g))
    print(f"--- B={b} S={s} C={c}: LN {t_ln:.1f}  qk {t_qk:.1f}  vT {t_vt:.1f}  to_q {t_q:.1f}  geglu {t_g:.1f} us", flush=True)
    for name, fn in (("qkv+ln+vt", lambda t: ops.linear_qkv(x, w_qkv, tile=t)), ("qkv+vt (no ln)", lambda t: ops.linear_qkv(x, w_qkv_noln, tile=t)),
                     ("to_q+ln", lambda t: ops.linear(x, w_q_ln, tile=t)), ("to_q ws-tile (no ln)", lambda t: ops.linear(n, w_q, tile=t)),
                     ("geglu+ln", lambda t: ops.linear_geglu(x, w_g_ln, tile=t)), ("geglu ws-tile (no ln)", lambda t: ops.linear_geglu(n, w_g, tile=t))):
        row = []
        for t in TILES:
            try:
                row.append(f"{t}:{bench(lambda: fn(t)):.1f}")
            except hip.MfhipError as e:
                row.append(f"{t}:n/a")
        print(f"   {name:24s} " + "  ".join(row), flush=True)
    ref = {"qkv+ln+vt": t_ln + t_qk + t_vt, "to_q+ln": t_ln + t_q, "geglu+ln": t_ln + t_g}
    print(f"   replaces: qkv {ref['qkv+ln+vt']:.1f}  to_q {ref['to_q+ln']:.1f}  geglu {ref['geglu+ln']:.1f} us", flush=True)
