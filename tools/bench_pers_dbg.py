"""Where does the time of the persistent 128-row GEMM (tile 70) go?  The same launch with its epilogue chunks, its MFMAs / fragment reads
and its DMAs switched off one at a time (MFHIP_DBG_EPI bits 1 / 2 / 4: results are garbage, the barrier structure is unchanged)."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from reflecting_reality_amd import hip, ops
    from bench_k import timed
    hip.AUTOTUNE = False
    dev = torch.device("cuda:0")
    prec = ops.Precision.get("bf16")
    g = torch.Generator().manual_seed(0)
    out = []
    for rows, k, n, kind in ((32768, 320, 1280, "plain"), (32768, 320, 2560, "geglu"), (32768, 320, 320, "res"), (8192, 640, 2560, "plain"), (2048, 1280, 5120, "plain")):
        x = torch.randn(rows, k, generator=g).to(dev, prec.act)
        w = torch.randn(n, k, generator=g) / k ** 0.5
        if kind == "geglu":
            gw = ops.geglu_weight(w, torch.zeros(n), prec, dev)
            us = timed(lambda: ops.linear_geglu(x, gw, tile=70))
        else:
            lw = ops.ConvWeight(w, torch.zeros(n), prec, dev)
            res = torch.randn(rows, n, generator=g).to(dev, prec.act) if kind == "res" else None
            us = timed(lambda: ops.linear(x, lw, tile=70, res0=res))
        out.append(f"{kind} {rows}x{k}->{n}: {us:7.1f}")
    print(" | ".join(out), flush=True)
    sys.exit(0)

for bits, what in ((0, "everything on"), (1, "no epilogue chunks"), (2, "no MFMA / fragment reads"), (4, "no DMA"), (5, "no epilogue, no DMA"), (6, "no MFMA, no DMA"), (7, "barriers only + dump"), (15, "barriers only, no dump"), (8, "everything but the dump"),
                   (16, "all on, compute waves at priority 1"), (32, "all on, staging waves at priority 1"), (64, "all on, epilogue waves at priority 1"), (48, "all on, compute + staging at priority 1")):
    env = dict(os.environ, MFHIP_DBG_EPI=str(bits))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env)
    print(f"MFHIP_DBG_EPI={bits} ({what}): {r.stdout.strip() or r.stderr[-300:]}", flush=True)
