"""Upper bound for fusing the 1x1 shortcut convs / BrushNet zero-convs into their consumers: time the 50-step
denoise with those launches REMOVED (results are wrong; timing only).  usage: exp_fuse_bound.py [sc] [zc]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import hip, ops, synth

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
pipe.set_progress_bar_config(disable=True)
inp = synth.pipeline_inputs(4, 512, 512)
noise = torch.randn(8, 4, 64, 64)
skip = set()
for m in (pipe.unet, pipe.brushnet):
    for k, w in m.P.items():
        if ("sc" in sys.argv and k.endswith("conv_shortcut")) or ("zc" in sys.argv and k.startswith("brushnet_")):
            skip.add(id(w))
print(f"skipping {len(skip)} weights", flush=True)
orig = ops.conv2d
bufs = {}


def patched(x, w, *a, **kw):
    if id(w) in skip:
        n = w.n if hasattr(w, "n") else w.cout
        key = (x.shape[:-1], n)
        if key not in bufs:
            bufs[key] = torch.zeros(*x.shape[:-1], n, device=x.device, dtype=x.dtype)
        return bufs[key]
    return orig(x, w, *a, **kw)


ops.conv2d = patched


def run():
    return pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
                image=inp["image"], mask=inp["mask"], depth=inp["depth"], num_inference_steps=50, guidance_scale=7.5,
                latents=inp["latents"], output_type="latent", height=512, width=512, conditioning_noise=noise).images


run(); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"args={sys.argv[1:]} whole call (latent out) best {best * 1e3:.1f} ms", flush=True)
