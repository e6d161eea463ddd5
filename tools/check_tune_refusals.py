"""Does the library refuse any (tile, split-K) the shipped tune cache names for the denoise step of BASELINE configs[1]?  (hip.gemm_conv forgets a
refused entry and retries with the heuristic tile — silently slower.)  Prints the refused keys; exit code 1 if there are any."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from reflecting_reality_amd import hip, synth  # noqa: E402

dev = torch.device("cuda", 0)
refused = []
orig = hip._tune_forget
hip._tune_forget = lambda ks: (refused.append((ks, hip.load().mf_last_error().decode())), orig(ks))[1]
for prec in sys.argv[1:] or ["bf16"]:
    pipe, _ = bench.build_pipeline(prec, dev)
    inp = {k: v.to(dev) for k, v in synth.pipeline_inputs(4, 512, 512, seed=1234, cross_dim=768).items()}
    pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"], mask=inp["mask"], depth=inp["depth"],
         num_inference_steps=3, guidance_scale=7.5, latents=inp["latents"], output_type="latent", brushnet_conditioning_scale=1.0, height=512, width=512,
         conditioning_noise=inp["vae_noise"])
    torch.cuda.synchronize()
    print(f"[{prec}] refused entries: {len(refused)}")
    for ks, why in refused:
        print("  ", ks, "--", why[:160])
    del pipe
sys.exit(1 if refused else 0)
