import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import synth
dev = torch.device("cuda", 0)
pipe, _ = bench.build_pipeline("bf16", dev)
inp = synth.pipeline_inputs(4, 512, 512)
for rep in range(2):
    for aux in (False, True):
        pipe.overlap_aux, pipe._graph_state = aux, None
        for _ in range(2):
            timing = {}
            pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"],
                 mask=inp["mask"], depth=inp["depth"], num_inference_steps=50, guidance_scale=7.5, latents=inp["latents"],
                 output_type="latent", height=512, width=512, conditioning_noise=inp["vae_noise"], _timing=timing)
            torch.cuda.synchronize()
        print(f"aux={aux}: {timing['denoise_start'].elapsed_time(timing['denoise_end']) / 50:.2f} ms per step", flush=True)
