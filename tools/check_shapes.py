"""Robustness sweep: the pipeline at other batch sizes / resolutions (shapes the tune cache has not seen)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import synth
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
for (b, h, w) in ((1, 512, 512), (2, 768, 512), (3, 384, 640), (8, 512, 512)):
    inp = synth.pipeline_inputs(b, h, w, seed=5)
    for rep in range(2):
        t0 = time.perf_counter()
        img = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"],
                   mask=inp["mask"], depth=inp["depth"], num_inference_steps=10, guidance_scale=7.5, latents=inp["latents"],
                   output_type="pt", height=h, width=w, conditioning_noise=inp["vae_noise"]).images
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert img.shape == (b, 3, h, w) and torch.isfinite(img).all()
    print(f"batch {b} {h}x{w}: ok, {dt * 1e3:.0f} ms for 10 steps + VAE ({dt * 100 / b:.1f} ms per image-step)", flush=True)
