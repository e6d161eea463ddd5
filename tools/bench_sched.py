"""Per-step time of the denoise loop for DDIM (hipGraph / eager), PNDM and UniPC (eager): batch 4 x 512x512."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import PNDMScheduler, UniPCMultistepScheduler, DDIMScheduler, synth

dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
inp = synth.pipeline_inputs(4, 512, 512)
base_cfg = pipe.scheduler.config


def run(name, sched, graph):
    pipe.scheduler, pipe.use_hip_graph, pipe._graph_state = sched, graph, None
    for _ in range(2):
        timing = {}
        pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"],
             mask=inp["mask"], depth=inp["depth"], num_inference_steps=50, guidance_scale=7.5, latents=inp["latents"],
             output_type="latent", height=512, width=512, conditioning_noise=inp["vae_noise"], _timing=timing)
        torch.cuda.synchronize()
    ms = timing["denoise_start"].elapsed_time(timing["denoise_end"])
    n = len(pipe.scheduler.timesteps)
    print(f"{name:32s} {ms / n:7.2f} ms per model evaluation ({n} evaluations, {ms:.0f} ms)", flush=True)


run("DDIM, captured hipGraph", DDIMScheduler.from_config(base_cfg), True)
run("DDIM, eager loop", DDIMScheduler.from_config(base_cfg), False)
run("PNDM (PLMS), graph + eager step", PNDMScheduler.from_config(base_cfg, skip_prk_steps=True), True)
run("PNDM (PLMS), eager loop", PNDMScheduler.from_config(base_cfg, skip_prk_steps=True), False)
run("UniPC, graph + eager step", UniPCMultistepScheduler.from_config(base_cfg), True)
run("UniPC, eager loop", UniPCMultistepScheduler.from_config(base_cfg), False)
