"""bf16 error envelope of the REFERENCE ITSELF: run the imported reference with bf16 weights / activations (what its
scripts do with torch_dtype=torch.bfloat16) on the same seeded cases as tools/make_golden.py and record how far its
results move from its own fp32 results (the committed goldens).  tests/ assert that the HIP bf16 path stays inside a
stated multiple of this envelope — a bound derived from the reference, not from our own kernels.

Runs only in the build container (needs /root/reference).  Output: tests/golden/bf16_envelope.json (numbers only).

    python tools/make_bf16_envelope.py            # tiny cases (seconds)
    python tools/make_bf16_envelope.py --full     # + SD1.5-size single step and BASELINE configs[0] (minutes)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as MG  # noqa: E402  (sets up the reference import path + shim)
from make_golden import (DDIMScheduler, PNDMScheduler, UniPCMultistepScheduler, StableDiffusionBrushNetPipeline,  # noqa: E402
                         BrushNetModel, R, synth, GOLD)
import diffusers.models.autoencoders.vae as ref_vae  # noqa: E402

BF = torch.bfloat16          # --dtype fp16 switches every cast below to torch.float16 (the reference scripts' default precision)
DT_NAME = "bf16"
torch.set_grad_enabled(False)


def stats(got, ref):
    got, ref = torch.as_tensor(got).float(), torch.as_tensor(ref).float()
    e = (got - ref).abs()
    return dict(linf=float(e.max()), mean=float(e.mean()), ref_absmax=float(ref.abs().max()), ref_absmean=float(ref.abs().mean()))


class fixed_noise:
    """The reference draws the VAE posterior noise with randn_tensor in the model dtype from the global RNG; the bf16
    stream differs from the fp32 one, so hand it the fp32 goldens' noise (cast) — the arithmetic stays the reference's."""

    def __init__(self, noises):
        self.noises = list(noises)

    def __enter__(self):
        self.orig = ref_vae.randn_tensor

        def fake(shape, generator=None, device=None, dtype=None, layout=None):
            n = self.noises.pop(0)
            assert tuple(n.shape) == tuple(shape), (n.shape, shape)
            return n.to(dtype)
        ref_vae.randn_tensor = fake
        return self

    def __exit__(self, *a):
        ref_vae.randn_tensor = self.orig


def sample_like(t, stride, n):
    return t.float().double().flatten()[::int(stride)][:n].float()


def tiny(out):
    ucfg, vcfg = R.TINY_UNET, R.TINY_VAE
    (unet, _, _), (brushnet, _, _), (vae, _, _) = MG.models(ucfg, vcfg, 0)
    for m in (unet, brushnet, vae):
        m.to(BF)
    G = np.load(os.path.join(GOLD, "tiny_models.npz"))
    g = torch.Generator().manual_seed(42)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 6, 8, 8, generator=g)
    ehs = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g)
    down, mid, up = brushnet(x.to(BF), 501, encoder_hidden_states=ehs.to(BF), brushnet_cond=cond.to(BF),
                             conditioning_scale=0.8, return_dict=False)
    for i, d in enumerate(down):
        out[f"tiny/bn_down_{i}"] = stats(d, G[f"bn_down_{i}"])
    out["tiny/bn_mid"] = stats(mid, G["bn_mid"])
    for i, u in enumerate(up):
        out[f"tiny/bn_up_{i}"] = stats(u, G[f"bn_up_{i}"])
    gd = [torch.from_numpy(G[f"bn_down_{i}"]).to(BF) for i in range(len(down))]
    gu = [torch.from_numpy(G[f"bn_up_{i}"]).to(BF) for i in range(len(up))]
    eps = unet(x.to(BF), 501, encoder_hidden_states=ehs.to(BF), down_block_add_samples=gd,
               mid_block_add_sample=torch.from_numpy(G["bn_mid"]).to(BF), up_block_add_samples=gu, return_dict=False)[0]
    out["tiny/unet_eps_inj"] = stats(eps, G["unet_eps_inj"])
    eps2 = unet(x.to(BF), 501, encoder_hidden_states=ehs.to(BF), down_block_add_samples=list(down), mid_block_add_sample=mid,
                up_block_add_samples=list(up), return_dict=False)[0]
    out["tiny/unet_eps_inj_chained"] = stats(eps2, G["unet_eps_inj"])
    out["tiny/unet_eps_plain"] = stats(unet(x.to(BF), 501, encoder_hidden_states=ehs.to(BF), return_dict=False)[0], G["unet_eps_plain"])
    img = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    out["tiny/vae_moments"] = stats(vae.encode(img.to(BF)).latent_dist.parameters, G["vae_moments"])
    z = torch.randn(2, 4, 8, 8, generator=g)
    out["tiny/vae_decode"] = stats(vae.decode(z.to(BF), return_dict=False)[0], G["vae_decode"])

    P = np.load(os.path.join(GOLD, "tiny_pipeline.npz"))
    sc = R.SD15_SCHED
    for name, cls, kw in (("ddim", DDIMScheduler, dict(clip_sample=False, set_alpha_to_one=False, steps_offset=1)),
                          ("pndm", PNDMScheduler, dict(skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1)),
                          ("unipc", PNDMScheduler, dict(skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1))):
        sched = cls(num_train_timesteps=1000, beta_start=sc["beta_start"], beta_end=sc["beta_end"],
                    beta_schedule="scaled_linear", **kw)
        if name == "unipc":
            sched = UniPCMultistepScheduler.from_config(sched.config)
        pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                               scheduler=sched, safety_checker=None, feature_extractor=None,
                                               requires_safety_checker=False, depth_conditioning_mode="concat")
        pipe.set_progress_bar_config(disable=True)
        inp = synth.pipeline_inputs(1, 16, 16, seed=1234, cross_dim=ucfg["cross_attention_dim"], vae_scale=2)
        trace, captured = [], {}
        hook = brushnet.register_forward_pre_hook(
            lambda mod, args, kwargs: captured.update(cond=kwargs["brushnet_cond"].clone()), with_kwargs=True)
        with fixed_noise([torch.from_numpy(P[f"{name}_vae_noise"])]):
            res = pipe(prompt_embeds=inp["prompt_embeds"].to(BF), negative_prompt_embeds=inp["negative_prompt_embeds"].to(BF),
                       image=inp["image"], mask=inp["mask"], depth=inp["depth"], num_inference_steps=4, guidance_scale=7.5,
                       latents=inp["latents"].clone().to(BF), output_type="pt", brushnet_conditioning_scale=1.0,
                       callback_on_step_end=lambda p, i, t, k: trace.append(k["latents"].clone()) or {}, height=16, width=16)
        hook.remove()
        out[f"tiny_pipeline/{name}/cond"] = stats(captured["cond"], P[f"{name}_cond"])
        for i, l in enumerate(trace):
            out[f"tiny_pipeline/{name}/latents_{i}"] = stats(l, P[f"{name}_latents_{i}"])
        out[f"tiny_pipeline/{name}/image"] = stats(res.images, P[f"{name}_image"])

    # alt conditioning modes (12-channel BrushNet)
    (unet32, _, _), _, _ = MG.models(ucfg, vcfg, 0)
    bn12 = BrushNetModel.from_unet(unet32, conditioning_channels=12, load_weights_from_unet=False).eval()
    MG.load_synth(bn12, 11)
    bn12.to(BF)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=sc["beta_start"], beta_end=sc["beta_end"],
                          beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=bn12,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="latents",
                                           normals_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 16, 16, seed=1234, cross_dim=ucfg["cross_attention_dim"], vae_scale=2)
    normals = torch.rand(1, 3, 16, 16, generator=torch.Generator().manual_seed(4321)) * 2.0 - 1.0
    captured = {}
    hook = bn12.register_forward_pre_hook(
        lambda mod, args, kwargs: captured.update(cond=kwargs["brushnet_cond"].clone()), with_kwargs=True)
    with fixed_noise([torch.from_numpy(P["alt_vae_noise"]), torch.from_numpy(P["alt_depth_noise"])]):
        res = pipe(prompt_embeds=inp["prompt_embeds"].to(BF), negative_prompt_embeds=inp["negative_prompt_embeds"].to(BF),
                   image=inp["image"], mask=inp["mask"], depth=inp["depth"], normals=normals, num_inference_steps=2,
                   guidance_scale=7.5, latents=inp["latents"].clone().to(BF), output_type="latent",
                   brushnet_conditioning_scale=1.0, height=16, width=16)
    hook.remove()
    out["tiny_pipeline/alt/cond"] = stats(captured["cond"], P["alt_cond"])
    out["tiny_pipeline/alt/latents"] = stats(res.images, P["alt_latents"])


def tiny_xl(out):
    from diffusers.pipelines.brushnet.pipeline_brushnet_sd_xl import StableDiffusionXLBrushNetPipeline as XLPipe
    ucfg, vcfg = R.TINY_XL_UNET, R.TINY_VAE
    unet = MG.build_unet(ucfg)
    MG.load_synth(unet, 20)
    brushnet = BrushNetModel.from_unet(unet, conditioning_channels=5, load_weights_from_unet=False).eval()
    MG.load_synth(brushnet, 21)
    vae = MG.build_vae(vcfg)
    MG.load_synth(vae, 2)
    for m in (unet, brushnet, vae):
        m.to(BF)
    G = np.load(os.path.join(GOLD, "tiny_xl.npz"))
    g = torch.Generator().manual_seed(43)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 5, 8, 8, generator=g)
    ehs = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g)
    added = dict(text_embeds=torch.randn(2, 24, generator=g).to(BF),
                 time_ids=torch.tensor([[16., 16., 0., 0., 16., 16.], [32., 24., 4., 2., 16., 16.]]).to(BF))
    down, mid, up = brushnet(x.to(BF), 401, encoder_hidden_states=ehs.to(BF), brushnet_cond=cond.to(BF), conditioning_scale=0.9,
                             added_cond_kwargs=added, return_dict=False)
    for i, d in enumerate(down):
        out[f"tiny_xl/bn_down_{i}"] = stats(d, G[f"bn_down_{i}"])
    out["tiny_xl/bn_mid"] = stats(mid, G["bn_mid"])
    for i, u in enumerate(up):
        out[f"tiny_xl/bn_up_{i}"] = stats(u, G[f"bn_up_{i}"])
    eps = unet(x.to(BF), 401, encoder_hidden_states=ehs.to(BF), added_cond_kwargs=added, down_block_add_samples=list(down),
               mid_block_add_sample=mid, up_block_add_samples=list(up), return_dict=False)[0]
    out["tiny_xl/unet_eps_inj"] = stats(eps, G["unet_eps_inj"])
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = XLPipe(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=unet,
                  brushnet=brushnet, scheduler=sched, force_zeros_for_empty_prompt=True, add_watermarker=False)
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 16, 16, seed=99, cross_dim=ucfg["cross_attention_dim"], vae_scale=2)
    gp = torch.Generator().manual_seed(100)
    pooled, npooled = torch.randn(1, 24, generator=gp), torch.randn(1, 24, generator=gp)
    with fixed_noise([torch.from_numpy(G["pipe_vae_noise"])]):
        res = pipe(prompt_embeds=inp["prompt_embeds"].to(BF), negative_prompt_embeds=inp["negative_prompt_embeds"].to(BF),
                   pooled_prompt_embeds=pooled.to(BF), negative_pooled_prompt_embeds=npooled.to(BF), image=inp["image"],
                   mask=inp["mask"], num_inference_steps=3, guidance_scale=5.0, latents=inp["latents"].clone().to(BF),
                   output_type="latent", brushnet_conditioning_scale=1.0, height=16, width=16, original_size=(24, 20),
                   crops_coords_top_left=(2, 1), target_size=(16, 16))
    out["tiny_xl/pipe_latents"] = stats(res.images, G["pipe_latents"])


def full(out):
    ucfg, vcfg = R.SD15_UNET, R.SD15_VAE
    (unet, _, _), (brushnet, _, _), (vae, _, _) = MG.models(ucfg, vcfg, 0)
    for m in (unet, brushnet, vae):
        m.to(BF)
    G = np.load(os.path.join(GOLD, "sd15_step.npz"))
    g = torch.Generator().manual_seed(43)
    lat = torch.randn(1, 4, 32, 32, generator=g)
    cond = torch.randn(2, 6, 32, 32, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g)
    x2 = torch.cat([lat] * 2).to(BF)
    down, mid, up = brushnet(x2, 981, encoder_hidden_states=ehs.to(BF), brushnet_cond=cond.to(BF), conditioning_scale=1.0,
                             return_dict=False)
    for name, ts in (("bn_down", down), ("bn_up", up)):
        for i, t in enumerate(ts):
            out[f"sd15_step/{name}_{i}"] = stats(sample_like(t, G[f"{name}_{i}_stats"][2], 256), G[f"{name}_{i}_sample"])
    out["sd15_step/bn_mid"] = stats(sample_like(mid, G["bn_mid_stats"][2], 256), G["bn_mid_sample"])
    eps = unet(x2, 981, encoder_hidden_states=ehs.to(BF), down_block_add_samples=list(down), mid_block_add_sample=mid,
               up_block_add_samples=list(up), return_dict=False)[0]
    out["sd15_step/eps"] = stats(eps, G["eps"])
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sched.set_timesteps(50)
    eu, ec = eps.chunk(2)
    lat1 = sched.step(eu + 7.5 * (ec - eu), 981, lat.to(BF), return_dict=False)[0]
    out["sd15_step/latents_after_step"] = stats(lat1, G["latents_after_step"])
    z = torch.randn(1, 4, 16, 16, generator=g)
    dec = vae.decode((z / vcfg["scaling_factor"]).to(BF), return_dict=False)[0]
    out["sd15_step/vae_decode"] = stats(sample_like(dec, G["vae_dec_stats"][2], 1024), G["vae_dec_sample"])
    img = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
    out["sd15_step/vae_moments"] = stats(vae.encode(img.to(BF)).latent_dist.parameters, G["vae_moments"])
    print("sd15 single step done", flush=True)

    C = np.load(os.path.join(GOLD, "sd15_config0.npz"))
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 256, 256, seed=1234)
    trace = []
    with fixed_noise([torch.from_numpy(C["vae_noise"])]):
        res = pipe(prompt_embeds=inp["prompt_embeds"].to(BF), negative_prompt_embeds=inp["negative_prompt_embeds"].to(BF),
                   image=inp["image"], mask=inp["mask"], depth=inp["depth"], num_inference_steps=4, guidance_scale=7.5,
                   latents=inp["latents"].clone().to(BF), output_type="pt", brushnet_conditioning_scale=1.0,
                   callback_on_step_end=lambda p, i, t, k: trace.append(k["latents"].clone()) or {}, height=256, width=256)
    for i, l in enumerate(trace):
        out[f"sd15_config0/latents_{i}"] = stats(l, C[f"latents_{i}"])
    out["sd15_config0/image"] = stats(sample_like(res.images, C["image_stats"][2], 1024), C["image_sample"])


def config1_slice(out):
    ucfg, vcfg = R.SD15_UNET, R.SD15_VAE
    (unet, _, _), (brushnet, _, _), (vae, _, _) = MG.models(ucfg, vcfg, 0)
    for m in (unet, brushnet, vae):
        m.to(BF)
    C = np.load(os.path.join(GOLD, "sd15_config1_slice.npz"))
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(4, 512, 512, seed=77)
    sl = slice(0, 1)
    nz = inp["vae_noise"]
    noise = torch.cat([nz[:4][sl], nz[4:][sl]])
    trace = []
    with fixed_noise([noise]):
        res = pipe(prompt_embeds=inp["prompt_embeds"][sl].to(BF), negative_prompt_embeds=inp["negative_prompt_embeds"][sl].to(BF),
                   image=inp["image"][sl], mask=inp["mask"][sl], depth=inp["depth"][sl], num_inference_steps=3,
                   guidance_scale=7.5, latents=inp["latents"][sl].clone().to(BF), output_type="pt",
                   brushnet_conditioning_scale=1.0,
                   callback_on_step_end=lambda p, i, t, k: trace.append(k["latents"].clone()) or {}, height=512, width=512)
    for i, l in enumerate(trace):
        out[f"sd15_config1_slice/latents_{i}"] = stats(l, C[f"latents_{i}"])
    out["sd15_config1_slice/image"] = stats(sample_like(res.images, C["image_stats"][2], 4096), C["image_sample"])


def config1_50steps(out):
    """The reference in bf16 over the benchmark's real 50 DDIM steps (the case of make_golden.config1_50steps)."""
    import time
    ucfg, vcfg = R.SD15_UNET, R.SD15_VAE
    (unet, _, _), (brushnet, _, _), (vae, _, _) = MG.models(ucfg, vcfg, 0)
    for m in (unet, brushnet, vae):
        m.to(BF)
    C = np.load(os.path.join(GOLD, "sd15_config1_50steps.npz"))
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(4, 512, 512, seed=77)
    sl = slice(0, 1)
    nz = inp["vae_noise"]
    noise = torch.cat([nz[:4][sl], nz[4:][sl]])
    trace = []
    t0 = time.time()

    def cb(p, i, t, k):
        trace.append(k["latents"].clone())
        print(f"[bf16 config1 50] step {i + 1} at {time.time() - t0:.0f} s", flush=True)
        return {}

    with fixed_noise([noise]):
        res = pipe(prompt_embeds=inp["prompt_embeds"][sl].to(BF), negative_prompt_embeds=inp["negative_prompt_embeds"][sl].to(BF),
                   image=inp["image"][sl], mask=inp["mask"][sl], depth=inp["depth"][sl], num_inference_steps=50,
                   guidance_scale=7.5, latents=inp["latents"][sl].clone().to(BF), output_type="pt",
                   brushnet_conditioning_scale=1.0, callback_on_step_end=cb, height=512, width=512)
    for n in MG.C1_50_STEPS:
        out[f"sd15_config1_50steps/latents_{n}"] = stats(trace[n - 1], C[f"latents_{n}"])
    out["sd15_config1_50steps/image"] = stats(sample_like(res.images, C["image_stats"][2], 4096), C["image_sample"])


def sdxl_full(out):
    """The reference in bf16 on the configs[4] slice (make_golden.sdxl_full)."""
    from diffusers.pipelines.brushnet.pipeline_brushnet_sd_xl import StableDiffusionXLBrushNetPipeline as XLPipe
    unet, brushnet, vae = MG._sdxl_full_models(BF)
    C = np.load(os.path.join(GOLD, "sdxl_config4_slice.npz"))
    trace, _ = MG.run_sdxl_full(XLPipe, unet, brushnet, vae, cast=BF)
    for i, l in enumerate(trace):
        out[f"sdxl_config4_slice/latents_{i}"] = stats(l, C[f"latents_{i}"])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-config1", action="store_true")
    ap.add_argument("--only-config1-50", action="store_true")
    ap.add_argument("--only-sdxl-full", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only-full", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"],
                    help="fp16: the reference's default torch_dtype (test_brushnet.py:122-126) -> tests/golden/fp16_envelope.json")
    a = ap.parse_args()
    if a.dtype == "fp16":
        BF, DT_NAME = torch.float16, "fp16"
    path = os.path.join(GOLD, f"{DT_NAME}_envelope.json")
    out = {}
    if os.path.exists(path):
        with open(path) as f:
            out = json.load(f)
    out["_about"] = (f"|reference({DT_NAME}) - reference(fp32)| of the imported reference on the golden cases "
                     "(tools/make_bf16_envelope.py): linf / mean abs error, and the fp32 result's abs max / mean")
    if a.only_sdxl_full:
        sdxl_full(out)
    elif a.only_config1_50:
        config1_50steps(out)
    elif a.only_config1:
        config1_slice(out)
    elif not a.only_full:
        tiny(out)
        tiny_xl(out)
    if (a.full or a.only_full) and not (a.only_config1 or a.only_config1_50 or a.only_sdxl_full):
        full(out)
        config1_slice(out)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        if k != "_about":
            print(f"{k:45s} linf {v['linf']:.3e} mean {v['mean']:.3e}  |ref| max {v['ref_absmax']:.2f} mean {v['ref_absmean']:.3f}")
