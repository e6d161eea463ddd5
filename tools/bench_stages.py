"""Stage timing of one pipeline call (batch 4 x 512x512): conditioning build (VAE encode), denoise loop, VAE decode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import hip, synth

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
pipe.set_progress_bar_config(disable=True)
inp = synth.pipeline_inputs(4, 512, 512)
noise = torch.randn(8, 4, 64, 64)


def sync_time(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


t_cond, cond = sync_time(lambda: pipe.build_conditioning(inp["image"], inp["mask"], inp["depth"], 512, 512, 4, 1, True, noise))
lat = inp["latents"].to(dev)
t_dec, _ = sync_time(lambda: pipe.vae.decode(lat, return_dict=False)[0])
t_all, _ = sync_time(lambda: pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
                                  image=inp["image"], mask=inp["mask"], depth=inp["depth"], num_inference_steps=50,
                                  guidance_scale=7.5, latents=inp["latents"], output_type="pt", height=512, width=512,
                                  conditioning_noise=noise).images, n=2)
print(f"conditioning build {t_cond:.1f} ms | vae.decode {t_dec:.1f} ms | whole call {t_all:.1f} ms", flush=True)
hip.tune_save()
