"""Full-size SDXL + BrushNet-XL (1024 x 1024, batch 1, 10 DDIM steps, seeded random weights): how far the fp8-Linear mode
and the bf16 mode move the final latents from the f16x3 parity mode on the same inputs.  One number per mode, for
profiles/: evidence that the fp8 path behaves at production width (its parity bound is asserted on the tiny-XL fixture)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reflecting_reality_amd import synth

dev = torch.device("cuda", 0)
B, S, STEPS = 1, 1024, 10
inp = synth.pipeline_inputs(B, S, S, seed=777, cross_dim=2048)
gp = torch.Generator().manual_seed(778)
pooled, npooled = torch.randn(B, 1280, generator=gp), torch.randn(B, 1280, generator=gp)
out = {}
ref = None
for prec in ("f16x3", "bf16", "fp8"):
    pipe, _ = bench.build_pipeline(prec, dev, model="sdxl")
    t0 = time.time()
    lat = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
               pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled, image=inp["image"], mask=inp["mask"],
               num_inference_steps=STEPS, guidance_scale=7.5, latents=inp["latents"].clone(), output_type="latent",
               brushnet_conditioning_scale=1.0, height=S, width=S, conditioning_noise=inp["vae_noise"]).images.float().cpu()
    if ref is None:
        ref = lat
        out["reference_mode"] = {"precision": prec, "latent_absmax": float(ref.abs().max()), "latent_absmean": float(ref.abs().mean())}
    else:
        e = (lat - ref).abs()
        out[prec] = {"linf": float(e.max()), "mean": float(e.mean()), "rel_l2": float((lat - ref).norm() / ref.norm())}
    print(prec, f"{time.time() - t0:.1f}s", out.get(prec, out["reference_mode"]), flush=True)
    del pipe
    torch.cuda.empty_cache()
out["workload"] = f"SDXL-base + BrushNet-XL shapes, seeded random weights, batch {B} x {S}x{S}, {STEPS} DDIM steps, CFG 7.5: final latents vs the f16x3 mode"
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r02_sdxl_mode_agreement.json", "w"), indent=1)
