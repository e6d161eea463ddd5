"""Generate the golden fixtures under tests/golden/ by running the IMPORTED REFERENCE.

Runs only in the build container (needs /root/reference); nothing under tests/, bench.py or the
package reads /root/reference at run time.  The fixtures are data: seeded inputs are regenerated from
`reflecting_reality_amd.synth`, only the reference's OUTPUTS (and key/shape tables) are stored.

    python tools/make_golden.py            # tiny fixtures (seconds)
    python tools/make_golden.py --full     # + full-size SD1.5-shape single-step fixtures (~2 min, ~12 GB RAM)

While generating it also checks the oracle restatement (oracle/mirrorfusion_ref.py) against the
reference on every case and prints the max abs difference.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/MirrorFusion/src")
import transformers.utils as _tu  # noqa: E402

if not hasattr(_tu, "FLAX_WEIGHTS_NAME"):          # removed in transformers 5; the reference imports it
    _tu.FLAX_WEIGHTS_NAME = "flax_model.msgpack"

from diffusers import (AutoencoderKL, BrushNetModel, DDIMScheduler, PNDMScheduler, UNet2DConditionModel,  # noqa: E402
                       UniPCMultistepScheduler)
from diffusers.pipelines.brushnet.pipeline_brushnet import StableDiffusionBrushNetPipeline  # noqa: E402

from oracle import mirrorfusion_ref as R  # noqa: E402
from reflecting_reality_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


def build_unet(cfg):
    extra = {k: cfg[k] for k in ("transformer_layers_per_block", "use_linear_projection", "addition_embed_type",
                                 "addition_time_embed_dim", "projection_class_embeddings_input_dim") if k in cfg}
    return UNet2DConditionModel(
        sample_size=8, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
        down_block_types=cfg["down_block_types"], up_block_types=cfg["up_block_types"],
        block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"],
        cross_attention_dim=cfg["cross_attention_dim"], attention_head_dim=cfg["attention_head_dim"],
        norm_num_groups=cfg["norm_num_groups"], **extra).eval()


def build_vae(cfg):
    n = len(cfg["block_out_channels"])
    return AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * n,
                         up_block_types=("UpDecoderBlock2D",) * n, block_out_channels=cfg["block_out_channels"],
                         layers_per_block=cfg["layers_per_block"], latent_channels=cfg["latent_channels"],
                         norm_num_groups=cfg["norm_num_groups"], scaling_factor=cfg["scaling_factor"]).eval()


def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.state_dict_for(shapes, seed)
    module.load_state_dict(sd, strict=True)
    return sd, shapes


def maxdiff(a, b):
    return float((a - b).abs().max())


def summarize(t: torch.Tensor, nsample: int = 256):
    """checksum + strided sample of a large tensor (full-size fixtures stay small)."""
    f = t.double().flatten()
    stride = max(1, f.numel() // nsample)
    return dict(shape=list(t.shape), sum=float(f.sum()), abssum=float(f.abs().sum()),
                sample_stride=stride, sample=f[::stride][:nsample].float().numpy())


def models(unet_cfg, vae_cfg, seed):
    unet = build_unet(unet_cfg)
    unet_sd, unet_shapes = load_synth(unet, seed)
    # load_weights_from_unet=True would alias unet.conv_in.bias into the BrushNet (brushnet.py:519), so loading
    # the BrushNet's synthetic weights would silently overwrite the UNet's bias
    brushnet = BrushNetModel.from_unet(unet, conditioning_channels=6, load_weights_from_unet=False).eval()
    bn_sd, bn_shapes = load_synth(brushnet, seed + 1)
    vae = build_vae(vae_cfg)
    vae_sd, vae_shapes = load_synth(vae, seed + 2)
    return (unet, unet_sd, unet_shapes), (brushnet, bn_sd, bn_shapes), (vae, vae_sd, vae_shapes)


def tiny():
    ucfg, vcfg = R.TINY_UNET, R.TINY_VAE
    (unet, unet_sd, unet_shapes), (brushnet, bn_sd, bn_shapes), (vae, vae_sd, vae_shapes) = models(ucfg, vcfg, 0)
    bcfg = R.brushnet_config(ucfg, 6)
    with open(os.path.join(GOLD, "keys_tiny.json"), "w") as f:
        json.dump(dict(unet=unet_shapes, brushnet=bn_shapes, vae=vae_shapes), f, indent=0, sort_keys=True)

    g = torch.Generator().manual_seed(42)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 6, 8, 8, generator=g)
    ehs = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g)
    t = 501
    out = {}
    # --- F3: BrushNet residuals -------------------------------------------------------------
    down, mid, up = brushnet(x, t, encoder_hidden_states=ehs, brushnet_cond=cond, conditioning_scale=0.8,
                             return_dict=False)
    od, om, ou = R.brushnet_forward(bn_sd, bcfg, x, t, cond, 0.8)
    print("brushnet oracle-vs-ref:", max(maxdiff(a, b) for a, b in zip(down + [mid] + up, od + [om] + ou)))
    for i, d in enumerate(down):
        out[f"bn_down_{i}"] = d.numpy()
    out["bn_mid"] = mid.numpy()
    for i, u in enumerate(up):
        out[f"bn_up_{i}"] = u.numpy()
    # --- F4: UNet with injection (the lists are consumed by pop(0): pass copies) --------------
    eps = unet(x, t, encoder_hidden_states=ehs, down_block_add_samples=list(down), mid_block_add_sample=mid,
               up_block_add_samples=list(up), return_dict=False)[0]
    oeps = R.unet_forward(unet_sd, ucfg, x, t, ehs, od, om, ou)
    print("unet+inj oracle-vs-ref:", maxdiff(eps, oeps))
    out["unet_eps_inj"] = eps.numpy()
    eps0 = unet(x, t, encoder_hidden_states=ehs, return_dict=False)[0]
    print("unet    oracle-vs-ref:", maxdiff(eps0, R.unet_forward(unet_sd, ucfg, x, t, ehs)))
    out["unet_eps_plain"] = eps0.numpy()
    # --- F8: VAE ---------------------------------------------------------------------------------
    img = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    mom = vae.encode(img).latent_dist.parameters
    print("vae enc oracle-vs-ref:", maxdiff(mom, R.vae_encode_moments(vae_sd, vcfg, img)))
    out["vae_moments"] = mom.numpy()
    z = torch.randn(2, 4, 8, 8, generator=g)
    dec = vae.decode(z, return_dict=False)[0]
    print("vae dec oracle-vs-ref:", maxdiff(dec, R.vae_decode(vae_sd, vcfg, z)))
    out["vae_decode"] = dec.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_models.npz"), **out)

    # --- F5: tiny pipeline, per-step latents for DDIM, PNDM and UniPC ------------------------------------
    pout = {}
    sched_cfg = {k: v for k, v in R.SD15_SCHED.items()}
    for name, cls, kw in (("ddim", DDIMScheduler, dict(clip_sample=False, set_alpha_to_one=False, steps_offset=1)),
                          ("pndm", PNDMScheduler, dict(skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1)),
                          ("unipc", PNDMScheduler, dict(skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1))):
        sched = cls(num_train_timesteps=1000, beta_start=sched_cfg["beta_start"], beta_end=sched_cfg["beta_end"],
                    beta_schedule="scaled_linear", **kw)
        if name == "unipc":          # examples/brushnet/test_brushnet.py:158
            sched = UniPCMultistepScheduler.from_config(sched.config)
        pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet,
                                               brushnet=brushnet, scheduler=sched, safety_checker=None,
                                               feature_extractor=None, requires_safety_checker=False,
                                               depth_conditioning_mode="concat")
        pipe.set_progress_bar_config(disable=True)
        inp = synth.pipeline_inputs(1, 16, 16, seed=1234, cross_dim=ucfg["cross_attention_dim"], vae_scale=2)
        captured = {}
        hook = brushnet.register_forward_pre_hook(
            lambda mod, args, kwargs: captured.update(cond=kwargs["brushnet_cond"].clone()), with_kwargs=True)
        trace = []

        def cb(p, i, t, kw_):
            trace.append(kw_["latents"].clone())
            return {}

        torch.manual_seed(777)       # the reference draws the VAE posterior noise from the global RNG (:1188)
        res = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
                   image=inp["image"], mask=inp["mask"], depth=inp["depth"], num_inference_steps=4,
                   guidance_scale=7.5, latents=inp["latents"].clone(), output_type="pt",
                   brushnet_conditioning_scale=1.0, callback_on_step_end=cb, height=16, width=16)
        hook.remove()
        torch.manual_seed(777)
        vae_noise = torch.randn(2, 4, 8, 8)
        # oracle replay with the explicit noise
        ocond = R.build_conditioning(vae_sd, vcfg, inp["image"], inp["mask"], inp["depth"], vae_noise)
        print(f"[{name}] conditioning oracle-vs-ref:", maxdiff(ocond, captured["cond"]))
        osched = {"ddim": R.DDIMRef, "pndm": R.PNDMRef, "unipc": R.UniPCRef}[name](**R.SD15_SCHED)
        otrace = []
        pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]])
        olat = R.denoise(unet_sd, ucfg, bn_sd, bcfg, osched, inp["latents"], ocond, pe, 4, 7.5, 1.0, otrace)
        print(f"[{name}] per-step latents oracle-vs-ref:", [round(maxdiff(a, b), 8) for a, b in zip(trace, otrace)])
        oimg = (R.vae_decode(vae_sd, vcfg, olat / vcfg["scaling_factor"]) / 2 + 0.5).clamp(0, 1)
        print(f"[{name}] image oracle-vs-ref:", maxdiff(oimg, res.images))
        pout[f"{name}_timesteps"] = pipe.scheduler.timesteps.numpy()
        pout[f"{name}_cond"] = captured["cond"].numpy()
        pout[f"{name}_vae_noise"] = vae_noise.numpy()
        for i, l in enumerate(trace):
            pout[f"{name}_latents_{i}"] = l.numpy()
        pout[f"{name}_image"] = res.images.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_pipeline.npz"), **pout)

    # --- F5b: the other conditioning modes (pipeline_brushnet.py:1203-1215): depth VAE-encoded ("latents", +4 ch)
    # and normals nearest-resized ("concat", +3 ch) -> a 12-channel BrushNet condition ---------------------------------
    bn12 = BrushNetModel.from_unet(unet, conditioning_channels=12, load_weights_from_unet=False).eval()
    bn12_sd, _ = load_synth(bn12, 11)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=sched_cfg["beta_start"], beta_end=sched_cfg["beta_end"],
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
    torch.manual_seed(777)
    res = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
               image=inp["image"], mask=inp["mask"], depth=inp["depth"], normals=normals, num_inference_steps=2,
               guidance_scale=7.5, latents=inp["latents"].clone(), output_type="latent",
               brushnet_conditioning_scale=1.0, height=16, width=16)
    hook.remove()
    torch.manual_seed(777)
    vae_noise = torch.randn(2, 4, 8, 8)
    depth_noise = torch.randn(2, 4, 8, 8)
    ocond = R.build_conditioning(vae_sd, vcfg, inp["image"], inp["mask"], inp["depth"], vae_noise, depth_mode="latents",
                                 depth_noise=depth_noise, normals=normals, normals_mode="concat")
    print("[alt modes] conditioning oracle-vs-ref:", maxdiff(ocond, captured["cond"]), tuple(ocond.shape))
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]])
    olat = R.denoise(unet_sd, ucfg, bn12_sd, R.brushnet_config(ucfg, 12), R.DDIMRef(**R.SD15_SCHED), inp["latents"], ocond,
                     pe, 2, 7.5, 1.0)
    print("[alt modes] 2-step latents oracle-vs-ref:", maxdiff(olat, res.images))
    pout["alt_cond"] = captured["cond"].numpy()
    pout["alt_vae_noise"] = vae_noise.numpy()
    pout["alt_depth_noise"] = depth_noise.numpy()
    pout["alt_latents"] = res.images.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_pipeline.npz"), **pout)

    # --- F1: scheduler traces with the SD1.5 constants -------------------------------------------
    sout = {}
    for n in (4, 50):
        d = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
        d.set_timesteps(n)
        sout[f"ddim_timesteps_{n}"] = d.timesteps.numpy()
        p = PNDMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1)
        p.set_timesteps(n)
        sout[f"pndm_timesteps_{n}"] = p.timesteps.numpy()
        # UniPC the way test_brushnet.py:158 builds it: from_config of the (PNDM) SD1.5 scheduler config
        u = UniPCMultistepScheduler.from_config(p.config)
        u.set_timesteps(n)
        sout[f"unipc_timesteps_{n}"] = u.timesteps.numpy()
        for nm, s in (("ddim", d), ("pndm", p), ("unipc", u)):
            gg = torch.Generator().manual_seed(5)
            x = torch.randn(2, 4, 8, 8, generator=gg)
            xs = []
            for t in s.timesteps:
                eps = torch.sin(x * 3.0 + float(t) * 0.01)          # deterministic stand-in model
                x = s.step(eps, t, x, return_dict=False)[0]
                xs.append(x.clone())
            sout[f"{nm}_trace_{n}"] = torch.stack(xs).numpy()
    sout["alphas_cumprod"] = d.alphas_cumprod.numpy()
    np.savez_compressed(os.path.join(GOLD, "schedulers.npz"), **sout)
    print("tiny fixtures written")


def tiny_train():
    """Forward half of the training step (examples/brushnet/train_brushnet_mirror.py:1407-1449).  The script itself
    is not importable here (autoroot / cudf / h5py), so the same library calls it makes are issued in its order:
    DDPMScheduler.add_noise / get_velocity, MirrorFusionModel.forward's BrushNet -> UNet composition (:858-888),
    diffusers.training_utils.compute_snr and F.mse_loss."""
    import torch.nn.functional as F
    from diffusers import DDPMScheduler
    from diffusers.training_utils import compute_snr
    ucfg, vcfg = R.TINY_UNET, R.TINY_VAE
    (unet, unet_sd, _), _, _ = models(ucfg, vcfg, 0)
    bn = BrushNetModel.from_unet(unet, conditioning_channels=5, load_weights_from_unet=False).eval()
    bn_sd, bn_shapes = load_synth(bn, 21)
    bcfg = R.brushnet_config(ucfg, 5)
    g = torch.Generator().manual_seed(2024)
    bsz = 3
    latents = torch.randn(bsz, 4, 8, 8, generator=g) * 0.8
    noise = torch.randn(bsz, 4, 8, 8, generator=g)
    cond = torch.randn(bsz, 5, 8, 8, generator=g)
    ehs = torch.randn(bsz, 77, ucfg["cross_attention_dim"], generator=g)
    timesteps = torch.tensor([17, 480, 965]).long()
    out = dict(timesteps=timesteps.numpy())
    with open(os.path.join(GOLD, "keys_tiny_train.json"), "w") as f:
        json.dump(dict(brushnet=bn_shapes), f, indent=0, sort_keys=True)
    for ptype in ("epsilon", "v_prediction"):
        sched_cfg = dict(R.SD15_SCHED, prediction_type=ptype)
        ns = DDPMScheduler(num_train_timesteps=1000, beta_start=sched_cfg["beta_start"], beta_end=sched_cfg["beta_end"],
                           beta_schedule="scaled_linear", prediction_type=ptype)
        noisy = ns.add_noise(latents, noise, timesteps)                                              # :1416
        down, mid, up = bn(noisy, timesteps, encoder_hidden_states=ehs, brushnet_cond=cond, return_dict=False)
        pred = unet(noisy, timesteps, encoder_hidden_states=ehs, down_block_add_samples=list(down),
                    mid_block_add_sample=mid, up_block_add_samples=list(up), return_dict=False)[0]
        target = noise if ptype == "epsilon" else ns.get_velocity(latents, noise, timesteps)         # :1427-1430
        out[f"{ptype}_noisy"] = noisy.numpy()
        out[f"{ptype}_pred"] = pred.numpy()
        for gamma in (None, 5.0):
            if gamma is None:
                loss = F.mse_loss(pred.float(), target.float(), reduction="mean")                    # :1434
            else:
                snr = compute_snr(ns, timesteps)                                                     # :1440-1449
                w = torch.stack([snr, gamma * torch.ones_like(timesteps)], dim=1).min(dim=1)[0]
                w = w / snr if ptype == "epsilon" else w / (snr + 1)
                loss = F.mse_loss(pred.float(), target.float(), reduction="none")
                loss = (loss.mean(dim=list(range(1, len(loss.shape)))) * w).mean()
            oloss, opred = R.training_loss(unet_sd, ucfg, bn_sd, bcfg, sched_cfg, latents, noise, timesteps, ehs, cond,
                                           gamma)
            print(f"[train {ptype} gamma={gamma}] loss ref={float(loss):.8f} oracle-vs-ref: loss",
                  abs(float(loss) - float(oloss)), "pred", maxdiff(pred, opred))
            out[f"{ptype}_loss_{'none' if gamma is None else 'snr5'}"] = np.float32(float(loss))
    np.savez_compressed(os.path.join(GOLD, "tiny_train.npz"), **out)
    print("tiny training-step fixture written")

    # ---- backward half (SURVEY.md §8 a-16 pin): two optimizer steps of the reference's own modules under torch
    # autograd — loss.backward(), clip_grad_norm_(1.0), torch.optim.AdamW(lr 1e-5, betas 0.9/0.999, wd 1e-2, eps 1e-8)
    # (train_brushnet_mirror.py:1188-1200,1459-1466) — BrushNet trainable, UNet frozen (the default) or trainable
    # (--train_base_unet).  Stored: loss and pre-clip gradient norm per step, the step-1 gradients of named tensors and
    # their total movement |w_2 - w_0| after the two steps. --------------------------------------------------------
    from diffusers import DDPMScheduler as _DDPM
    torch.set_grad_enabled(True)
    tout = {}
    named_bn = ["conv_in_condition.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.conv1.weight",
                "down_blocks.0.resnets.1.norm2.weight", "down_blocks.1.resnets.0.time_emb_proj.bias",
                "down_blocks.0.downsamplers.0.conv.weight", "mid_block.resnets.0.conv2.bias",
                "up_blocks.0.resnets.0.conv_shortcut.weight", "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.1.resnets.2.norm1.bias",
                "brushnet_down_blocks.0.weight", "brushnet_mid_block.bias", "brushnet_up_blocks.3.weight"]
    named_un = ["conv_in.weight", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight",
                "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_v.weight",
                "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
                "down_blocks.0.attentions.0.transformer_blocks.0.norm2.weight", "down_blocks.0.attentions.0.proj_out.bias",
                "mid_block.attentions.0.norm.weight", "up_blocks.1.attentions.2.transformer_blocks.0.attn1.to_out.0.bias",
                "up_blocks.1.resnets.0.conv1.weight", "conv_norm_out.bias", "conv_out.weight"]
    g2 = torch.Generator().manual_seed(2025)
    batches = [(latents, noise, timesteps, ehs, cond),
               (torch.randn(bsz, 4, 8, 8, generator=g2) * 0.8, torch.randn(bsz, 4, 8, 8, generator=g2),
                torch.tensor([702, 3, 250]).long(), torch.randn(bsz, 77, ucfg["cross_attention_dim"], generator=g2),
                torch.randn(bsz, 5, 8, 8, generator=g2))]
    for tag, train_unet, gamma in (("frozen", False, None), ("unet", True, 5.0)):
        (unet2, unet_sd2, _), _, _ = models(ucfg, vcfg, 0)
        bn2 = BrushNetModel.from_unet(unet2, conditioning_channels=5, load_weights_from_unet=False)
        bn_sd2, _ = load_synth(bn2, 21)
        bn2.train()
        unet2.requires_grad_(train_unet)
        if train_unet:
            unet2.train()
        plist = list(bn2.parameters()) + (list(unet2.parameters()) if train_unet else [])
        opt = torch.optim.AdamW(plist, lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
        ns = _DDPM(num_train_timesteps=1000, beta_start=R.SD15_SCHED["beta_start"], beta_end=R.SD15_SCHED["beta_end"],
                   beta_schedule="scaled_linear")
        w0 = {k: v.detach().clone() for k, v in bn2.state_dict().items()}
        u0 = {k: v.detach().clone() for k, v in unet2.state_dict().items()}
        for i, (lat_, noi_, ts_, ehs_, cond_) in enumerate(batches):
            noisy = ns.add_noise(lat_, noi_, ts_)
            down, mid, up = bn2(noisy, ts_, encoder_hidden_states=ehs_, brushnet_cond=cond_, return_dict=False)
            pred = unet2(noisy, ts_, encoder_hidden_states=ehs_, down_block_add_samples=list(down), mid_block_add_sample=mid,
                         up_block_add_samples=list(up), return_dict=False)[0]
            if gamma is None:
                loss = F.mse_loss(pred.float(), noi_.float(), reduction="mean")
            else:
                snr = compute_snr(ns, ts_)
                w = torch.stack([snr, gamma * torch.ones_like(ts_)], dim=1).min(dim=1)[0] / snr
                loss = (F.mse_loss(pred.float(), noi_.float(), reduction="none").mean(dim=[1, 2, 3]) * w).mean()
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(plist, 1.0)                       # accelerator.clip_grad_norm_ (:1463)
            tout[f"{tag}_loss_{i}"] = np.float32(float(loss))
            tout[f"{tag}_grad_norm_{i}"] = np.float32(float(gn))
            if i == 0:
                coef = min(1.0, 1.0 / (float(gn) + 1e-6))
                for k in named_bn:
                    tout[f"{tag}_grad/{k}"] = (dict(bn2.named_parameters())[k].grad / coef).detach().numpy().copy()
                if train_unet:
                    for k in named_un:
                        tout[f"{tag}_grad/unet.{k}"] = (dict(unet2.named_parameters())[k].grad / coef).detach().numpy().copy()
            opt.step()
            opt.zero_grad()
        for k in named_bn:
            tout[f"{tag}_dw/{k}"] = np.float32(float((bn2.state_dict()[k] - w0[k]).norm()))
        if train_unet:
            for k in named_un:
                tout[f"{tag}_dw/unet.{k}"] = np.float32(float((unet2.state_dict()[k] - u0[k]).norm()))
        # the oracle's autograd + hand-written AdamW on the same batches
        nb, nu, rec = R.training_steps(unet_sd2, ucfg, bn_sd2, bcfg, dict(R.SD15_SCHED), batches, lr=1e-5, snr_gamma=gamma,
                                       train_unet=train_unet)
        print(f"[train backward {tag}] loss", [r["loss"] for r in rec], "ref", [float(tout[f"{tag}_loss_{i}"]) for i in range(2)],
              "grad norm", [r["grad_norm"] for r in rec], "ref", [float(tout[f"{tag}_grad_norm_{i}"]) for i in range(2)])
        print(f"[train backward {tag}] oracle-vs-ref: max grad diff",
              max(maxdiff(rec[0]["grads"][k], torch.from_numpy(tout[f"{tag}_grad/{k}"])) for k in named_bn),
              "max weight diff after 2 steps", max(maxdiff(nb[k], bn2.state_dict()[k]) for k in nb),
              ("unet " + str(max(maxdiff(nu[k], unet2.state_dict()[k]) for k in nu))) if train_unet else "")
    torch.set_grad_enabled(False)
    np.savez_compressed(os.path.join(GOLD, "tiny_train_backward.npz"), **tout)
    print("tiny training backward fixture written")


def tiny_guess_mode():
    """guess_mode (brushnet.py:896-902, pipeline_brushnet.py:1260-1264,1287-1293): log-spaced residual scales, BrushNet on
    the conditional batch only, zeros for the unconditional half.  Model-level residuals + a 4-step DDIM pipeline run."""
    ucfg, vcfg = R.TINY_UNET, R.TINY_VAE
    (unet, unet_sd, _), (brushnet, bn_sd, _), (vae, vae_sd, _) = models(ucfg, vcfg, 0)
    bcfg = R.brushnet_config(ucfg, 6)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 6, 8, 8, generator=g)
    ehs = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g)
    out = {}
    down, mid, up = brushnet(x, 501, encoder_hidden_states=ehs, brushnet_cond=cond, conditioning_scale=0.8, guess_mode=True,
                             return_dict=False)
    od, om, ou = R.brushnet_forward(bn_sd, bcfg, x, 501, cond, 0.8, guess_mode=True)
    print("[guess] brushnet oracle-vs-ref:", max(maxdiff(a, b) for a, b in zip(list(down) + [mid] + list(up), od + [om] + ou)))
    for i, d in enumerate(down):
        out[f"bn_down_{i}"] = d.numpy()
    out["bn_mid"] = mid.numpy()
    for i, u in enumerate(up):
        out[f"bn_up_{i}"] = u.numpy()
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=R.SD15_SCHED["beta_start"], beta_end=R.SD15_SCHED["beta_end"],
                          beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(2, 16, 32, seed=4321, cross_dim=ucfg["cross_attention_dim"], vae_scale=2)
    trace = []

    def cb(p_, i, t, kw_):
        trace.append(kw_["latents"].clone())
        return {}

    torch.manual_seed(778)           # the reference draws the VAE posterior noise from the global RNG (:1188); not CFG-doubled here
    res = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"], image=inp["image"],
               mask=inp["mask"], depth=inp["depth"], num_inference_steps=4, guidance_scale=7.5, latents=inp["latents"].clone(),
               output_type="latent", brushnet_conditioning_scale=0.9, guess_mode=True, callback_on_step_end=cb, height=16, width=32)
    torch.manual_seed(778)
    vae_noise = torch.randn(2, 4, 8, 16)
    ocond = R.build_conditioning(vae_sd, vcfg, inp["image"], inp["mask"], inp["depth"], vae_noise, cfg_dup=False)
    otrace = []
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]])
    R.denoise(unet_sd, ucfg, bn_sd, bcfg, R.DDIMRef(**R.SD15_SCHED), inp["latents"], ocond, pe, 4, 7.5, 0.9, otrace, guess_mode=True)
    print("[guess] per-step latents oracle-vs-ref:", [round(maxdiff(a, b), 8) for a, b in zip(trace, otrace)])
    out["vae_noise"] = vae_noise.numpy()
    for i, l in enumerate(trace):
        out[f"latents_{i}"] = l.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_guess_mode.npz"), **out)
    print("guess-mode fixture written")


def layers_full():
    """F6: single layers at the production sizes of the SD1.5 path, from the reference's own modules
    (ResnetBlock2D 320->320 @64x64, ResnetBlock2D 2560->1280 @16x16 with the 1x1 shortcut, Transformer2DModel 320 ch /
    4096 tokens / 8 heads with 77x768 cross-attention, Attention self-attention S=4096 d=40); strided samples + sums."""
    from diffusers.models.attention_processor import Attention
    from diffusers.models.resnet import ResnetBlock2D
    from diffusers.models.transformers.transformer_2d import Transformer2DModel
    g = torch.Generator().manual_seed(606)
    out = {}
    shapes_all = {}

    def put(name, t):
        s = summarize(t)
        out[name + "_sample"] = s["sample"]
        out[name + "_stats"] = np.array([s["sum"], s["abssum"], s["sample_stride"]])

    for name, cin, cout, hw, seed in (("resnet_320_64", 320, 320, 64, 61), ("resnet_2560_1280_16", 2560, 1280, 16, 62)):
        m = ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=1280, groups=32, eps=1e-5).eval()
        sd, shapes = load_synth(m, seed)
        shapes_all[name] = shapes
        x = torch.randn(1, cin, hw, hw, generator=g)
        temb = torch.randn(1, 1280, generator=g)
        y = m(x, temb)
        print(f"[{name}] oracle-vs-ref:", maxdiff(y, R.resnet(sd, "", x, temb, 32, 1e-5)), "absmax", float(y.abs().max()))
        put(name, y)
    m = Transformer2DModel(num_attention_heads=8, attention_head_dim=40, in_channels=320, num_layers=1,
                           cross_attention_dim=768, norm_num_groups=32).eval()
    sd, shapes = load_synth(m, 63)
    shapes_all["transformer_320_4096"] = shapes
    x = torch.randn(1, 320, 64, 64, generator=g)
    ehs = torch.randn(1, 77, 768, generator=g)
    y = m(x, encoder_hidden_states=ehs, return_dict=False)[0]
    print("[transformer_320_4096] oracle-vs-ref:", maxdiff(y, R.transformer_2d(sd, "", x, ehs, 8, 32)), "absmax", float(y.abs().max()))
    put("transformer_320_4096", y)
    a = Attention(query_dim=320, heads=8, dim_head=40, bias=False).eval()
    sd, shapes = load_synth(a, 64)
    shapes_all["attention_4096_40"] = shapes
    tok = torch.randn(1, 4096, 320, generator=g)
    y = a(tok)
    print("[attention_4096_40] oracle-vs-ref:", maxdiff(y, R.attention(sd, "", tok, None, 8)), "absmax", float(y.abs().max()))
    put("attention_4096_40", y)
    with open(os.path.join(GOLD, "keys_sd15_layers.json"), "w") as f:
        json.dump(shapes_all, f, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "sd15_layers.npz"), **out)
    print("full-size layer fixtures written")


def tiny_xl():
    """SDXL architecture (linear projections, per-level depth / heads, text_time embedding) on a tiny configuration:
    BrushNet-XL residuals, UNet-XL with injection, and a 3-step StableDiffusionXLBrushNetPipeline run."""
    from diffusers.pipelines.brushnet.pipeline_brushnet_sd_xl import StableDiffusionXLBrushNetPipeline
    ucfg, vcfg = R.TINY_XL_UNET, R.TINY_VAE
    unet = build_unet(ucfg)
    unet_sd, unet_shapes = load_synth(unet, 20)
    brushnet = BrushNetModel.from_unet(unet, conditioning_channels=5, load_weights_from_unet=False).eval()
    bn_sd, bn_shapes = load_synth(brushnet, 21)
    vae = build_vae(vcfg)
    vae_sd, vae_shapes = load_synth(vae, 2)
    bcfg = R.brushnet_config(ucfg, 5)
    with open(os.path.join(GOLD, "keys_tiny_xl.json"), "w") as f:
        json.dump(dict(unet=unet_shapes, brushnet=bn_shapes, vae=vae_shapes), f, indent=0, sort_keys=True)
    g = torch.Generator().manual_seed(43)
    x = torch.randn(2, 4, 8, 8, generator=g)
    cond = torch.randn(2, 5, 8, 8, generator=g)
    ehs = torch.randn(2, 77, ucfg["cross_attention_dim"], generator=g)
    added = dict(text_embeds=torch.randn(2, 24, generator=g),
                 time_ids=torch.tensor([[16., 16., 0., 0., 16., 16.], [32., 24., 4., 2., 16., 16.]]))
    t = 401
    out = {}
    down, mid, up = brushnet(x, t, encoder_hidden_states=ehs, brushnet_cond=cond, conditioning_scale=0.9,
                             added_cond_kwargs=added, return_dict=False)
    od, om, ou = R.brushnet_forward(bn_sd, bcfg, x, t, cond, 0.9, added)
    print("[xl] brushnet oracle-vs-ref:", max(maxdiff(a, b) for a, b in zip(down + [mid] + up, od + [om] + ou)))
    for i, d in enumerate(down):
        out[f"bn_down_{i}"] = d.numpy()
    out["bn_mid"] = mid.numpy()
    for i, u in enumerate(up):
        out[f"bn_up_{i}"] = u.numpy()
    eps = unet(x, t, encoder_hidden_states=ehs, added_cond_kwargs=added, down_block_add_samples=list(down),
               mid_block_add_sample=mid, up_block_add_samples=list(up), return_dict=False)[0]
    oeps = R.unet_forward(unet_sd, ucfg, x, t, ehs, od, om, ou, added)
    print("[xl] unet+inj oracle-vs-ref:", maxdiff(eps, oeps))
    out["unet_eps_inj"] = eps.numpy()
    # --- pipeline: 3 DDIM steps, CFG 5.0, 16x16 image (latents 8x8) -------------------------------------------
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionXLBrushNetPipeline(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None,
                                             tokenizer_2=None, unet=unet, brushnet=brushnet, scheduler=sched,
                                             force_zeros_for_empty_prompt=True, add_watermarker=False)
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 16, 16, seed=99, cross_dim=ucfg["cross_attention_dim"], vae_scale=2)
    gp = torch.Generator().manual_seed(100)
    pooled, npooled = torch.randn(1, 24, generator=gp), torch.randn(1, 24, generator=gp)
    captured = {}
    hook = brushnet.register_forward_pre_hook(
        lambda mod, args, kwargs: captured.update(cond=kwargs["brushnet_cond"].clone(),
                                                  tid=kwargs["added_cond_kwargs"]["time_ids"].clone()), with_kwargs=True)
    torch.manual_seed(778)
    res = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
               pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled, image=inp["image"], mask=inp["mask"],
               num_inference_steps=3, guidance_scale=5.0, latents=inp["latents"].clone(), output_type="latent",
               brushnet_conditioning_scale=1.0, height=16, width=16, original_size=(24, 20), crops_coords_top_left=(2, 1),
               target_size=(16, 16))
    hook.remove()
    torch.manual_seed(778)
    vae_noise = torch.randn(2, 4, 8, 8)
    ocond = R.build_conditioning(vae_sd, vcfg, inp["image"], inp["mask"], None, vae_noise)
    print("[xl] conditioning oracle-vs-ref:", maxdiff(ocond, captured["cond"]), tuple(captured["tid"].shape), captured["tid"][0].tolist())
    tid = torch.tensor([[24., 20., 2., 1., 16., 16.]]).repeat(2, 1)
    oadded = dict(text_embeds=torch.cat([npooled, pooled]), time_ids=tid)
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]])
    olat = R.denoise(unet_sd, ucfg, bn_sd, bcfg, R.DDIMRef(**R.SD15_SCHED), inp["latents"], ocond, pe, 3, 5.0, 1.0, None, oadded)
    print("[xl] 3-step latents oracle-vs-ref:", maxdiff(olat, res.images))
    out["pipe_cond"] = captured["cond"].numpy()
    out["pipe_vae_noise"] = vae_noise.numpy()
    out["pipe_latents"] = res.images.numpy()
    np.savez_compressed(os.path.join(GOLD, "tiny_xl.npz"), **out)
    print("tiny xl fixtures written")


def full():
    """F6/F7: full-size SD1.5-shape single step at 32x32 latents (1 image with CFG) + VAE decode/encode."""
    ucfg, vcfg = R.SD15_UNET, R.SD15_VAE
    (unet, unet_sd, unet_shapes), (brushnet, bn_sd, bn_shapes), (vae, vae_sd, vae_shapes) = models(ucfg, vcfg, 0)
    bcfg = R.brushnet_config(ucfg, 6)
    with open(os.path.join(GOLD, "keys_sd15.json"), "w") as f:
        json.dump(dict(unet=unet_shapes, brushnet=bn_shapes, vae=vae_shapes), f, indent=0, sort_keys=True)
    g = torch.Generator().manual_seed(43)
    lat = torch.randn(1, 4, 32, 32, generator=g)
    cond = torch.randn(2, 6, 32, 32, generator=g)
    ehs = torch.randn(2, 77, 768, generator=g)
    t = 981
    x2 = torch.cat([lat] * 2)
    down, mid, up = brushnet(x2, t, encoder_hidden_states=ehs, brushnet_cond=cond, conditioning_scale=1.0,
                             return_dict=False)
    eps = unet(x2, t, encoder_hidden_states=ehs, down_block_add_samples=list(down), mid_block_add_sample=mid,
               up_block_add_samples=list(up), return_dict=False)[0]
    od, om, ou = R.brushnet_forward(bn_sd, bcfg, x2, t, cond, 1.0)
    oeps = R.unet_forward(unet_sd, ucfg, x2, t, ehs, od, om, ou)
    print("full brushnet oracle-vs-ref:", max(maxdiff(a, b) for a, b in zip(down + [mid] + up, od + [om] + ou)))
    print("full unet eps oracle-vs-ref:", maxdiff(eps, oeps), "eps absmax", float(eps.abs().max()))
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sched.set_timesteps(50)
    eu, ec = eps.chunk(2)
    guided = eu + 7.5 * (ec - eu)
    lat1 = sched.step(guided, t, lat, return_dict=False)[0]
    out = dict(eps=eps.numpy(), latents_after_step=lat1.numpy())
    for i, d in enumerate(down):
        s = summarize(d)
        out[f"bn_down_{i}_sample"] = s["sample"]
        out[f"bn_down_{i}_stats"] = np.array([s["sum"], s["abssum"], s["sample_stride"]])
    s = summarize(mid)
    out["bn_mid_sample"], out["bn_mid_stats"] = s["sample"], np.array([s["sum"], s["abssum"], s["sample_stride"]])
    for i, u in enumerate(up):
        s = summarize(u)
        out[f"bn_up_{i}_sample"] = s["sample"]
        out[f"bn_up_{i}_stats"] = np.array([s["sum"], s["abssum"], s["sample_stride"]])
    # VAE at 128x128 image (16x16 latents) keeps the fixture generation fast
    z = torch.randn(1, 4, 16, 16, generator=g)
    dec = vae.decode(z / vcfg["scaling_factor"], return_dict=False)[0]
    print("full vae dec oracle-vs-ref:", maxdiff(dec, R.vae_decode(vae_sd, vcfg, z / vcfg["scaling_factor"])))
    s = summarize(dec, 1024)
    out["vae_dec_sample"], out["vae_dec_stats"] = s["sample"], np.array([s["sum"], s["abssum"], s["sample_stride"]])
    img = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
    mom = vae.encode(img).latent_dist.parameters
    print("full vae enc oracle-vs-ref:", maxdiff(mom, R.vae_encode_moments(vae_sd, vcfg, img)))
    out["vae_moments"] = mom.numpy()
    np.savez_compressed(os.path.join(GOLD, "sd15_step.npz"), **out)
    print("full-size fixtures written")

    # --- BASELINE.json configs[0]: SD1.5 + BrushNet depth-cond inpaint, 1 x 256 x 256, 4 DDIM steps, CFG 7.5, fp32,
    # through the reference pipeline's __call__ (the reference's own CPU-runnable case) -------------------------------
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(1, 256, 256, seed=1234)
    trace = []

    def cb(p_, i, t_, kw_):
        trace.append(kw_["latents"].clone())
        return {}

    torch.manual_seed(777)                    # the VAE posterior noise comes from the global RNG (:1188)
    res = pipe(prompt_embeds=inp["prompt_embeds"], negative_prompt_embeds=inp["negative_prompt_embeds"],
               image=inp["image"], mask=inp["mask"], depth=inp["depth"], num_inference_steps=4, guidance_scale=7.5,
               latents=inp["latents"].clone(), output_type="pt", brushnet_conditioning_scale=1.0,
               callback_on_step_end=cb, height=256, width=256)
    torch.manual_seed(777)
    vae_noise = torch.randn(2, 4, 32, 32)
    ocond = R.build_conditioning(vae_sd, vcfg, inp["image"], inp["mask"], inp["depth"], vae_noise)
    otrace = []
    pe = torch.cat([inp["negative_prompt_embeds"], inp["prompt_embeds"]])
    olat = R.denoise(unet_sd, ucfg, bn_sd, bcfg, R.DDIMRef(**R.SD15_SCHED), inp["latents"], ocond, pe, 4, 7.5, 1.0, otrace)
    print("[config0] per-step latents oracle-vs-ref:", [round(maxdiff(a, b), 8) for a, b in zip(trace, otrace)])
    oimg = (R.vae_decode(vae_sd, vcfg, olat / vcfg["scaling_factor"]) / 2 + 0.5).clamp(0, 1)
    print("[config0] image oracle-vs-ref:", maxdiff(oimg, res.images))
    c0 = dict(vae_noise=vae_noise.numpy(), timesteps=pipe.scheduler.timesteps.numpy())
    for i, l in enumerate(trace):
        c0[f"latents_{i}"] = l.numpy()
    s_ = summarize(res.images, 1024)
    c0["image_sample"], c0["image_stats"] = s_["sample"], np.array([s_["sum"], s_["abssum"], s_["sample_stride"]])
    np.savez_compressed(os.path.join(GOLD, "sd15_config0.npz"), **c0)
    print("configs[0] fixture written")


def config1_slice():
    """BASELINE.json configs[1] sizes through the reference pipeline for ONE image of the batch: image 0 of the batch-4
    x 512 x 512 synthetic inputs (seed 77), 3 DDIM steps, CFG 7.5, decoded (64 x 64 latents: the VAE's 4096-token
    single-head d = 512 attention).  The GPU tests run the whole batch of 4 and compare image 0 with this."""
    ucfg, vcfg = R.SD15_UNET, R.SD15_VAE
    (unet, unet_sd, _), (brushnet, bn_sd, _), (vae, vae_sd, _) = models(ucfg, vcfg, 0)
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = StableDiffusionBrushNetPipeline(vae=vae, text_encoder=None, tokenizer=None, unet=unet, brushnet=brushnet,
                                           scheduler=sched, safety_checker=None, feature_extractor=None,
                                           requires_safety_checker=False, depth_conditioning_mode="concat")
    pipe.set_progress_bar_config(disable=True)
    inp = synth.pipeline_inputs(4, 512, 512, seed=77)
    sl = slice(0, 1)
    nz = inp["vae_noise"]
    noise = torch.cat([nz[:4][sl], nz[4:][sl]])                 # uncond half, then cond half (what the HIP test passes)
    import diffusers.models.autoencoders.vae as ref_vae
    orig = ref_vae.randn_tensor
    ref_vae.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: noise.to(dtype)
    trace = []
    try:
        res = pipe(prompt_embeds=inp["prompt_embeds"][sl], negative_prompt_embeds=inp["negative_prompt_embeds"][sl],
                   image=inp["image"][sl], mask=inp["mask"][sl], depth=inp["depth"][sl], num_inference_steps=3,
                   guidance_scale=7.5, latents=inp["latents"][sl].clone(), output_type="pt", brushnet_conditioning_scale=1.0,
                   callback_on_step_end=lambda p_, i, t_, kw_: trace.append(kw_["latents"].clone()) or {}, height=512, width=512)
    finally:
        ref_vae.randn_tensor = orig
    ocond = R.build_conditioning(vae_sd, vcfg, inp["image"][sl], inp["mask"][sl], inp["depth"][sl], noise)
    pe = torch.cat([inp["negative_prompt_embeds"][sl], inp["prompt_embeds"][sl]])
    otrace = []
    olat = R.denoise(unet_sd, ucfg, bn_sd, R.brushnet_config(ucfg, 6), R.DDIMRef(**R.SD15_SCHED), inp["latents"][sl], ocond, pe,
                     3, 7.5, 1.0, otrace)
    print("[config1 slice] per-step latents oracle-vs-ref:", [round(maxdiff(a, b), 8) for a, b in zip(trace, otrace)])
    oimg = (R.vae_decode(vae_sd, vcfg, olat / vcfg["scaling_factor"]) / 2 + 0.5).clamp(0, 1)
    print("[config1 slice] image oracle-vs-ref:", maxdiff(oimg, res.images))
    out = dict(timesteps=pipe.scheduler.timesteps.numpy())
    for i, l in enumerate(trace):
        out[f"latents_{i}"] = l.numpy()
    s_ = summarize(res.images, 4096)
    out["image_sample"], out["image_stats"] = s_["sample"], np.array([s_["sum"], s_["abssum"], s_["sample_stride"]])
    np.savez_compressed(os.path.join(GOLD, "sd15_config1_slice.npz"), **out)
    print("configs[1] slice fixture written")


C1_50_STEPS = (1, 5, 10, 20, 30, 40, 50)


def config1_50steps():
    """BASELINE.json configs[1] at the benchmark's REAL length: image 0 of the batch-4 x 512 x 512 synthetic inputs (seed 77,
    the inputs of config1_slice) through ALL 50 DDIM steps of the imported reference pipeline (pipeline_brushnet.py:1250-1332,
    scheduling_ddim.py:344-470), CFG 7.5, fp32.  Stored: the latents after steps 1, 5, 10, 20, 30, 40, 50 and a strided sample
    of the decoded image.  (The oracle is bit-exact with the reference on the first 3 steps of exactly this run —
    config1_slice — and is not re-run for 50.)"""
    import time
    ucfg, vcfg = R.SD15_UNET, R.SD15_VAE
    (unet, _, _), (brushnet, _, _), (vae, _, _) = models(ucfg, vcfg, 0)
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
    import diffusers.models.autoencoders.vae as ref_vae
    orig = ref_vae.randn_tensor
    ref_vae.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: noise.to(dtype)
    trace = []
    t0 = time.time()

    def cb(p_, i, t_, kw_):
        trace.append(kw_["latents"].clone())
        print(f"[config1 50] step {i + 1} at {time.time() - t0:.0f} s", flush=True)
        return {}

    try:
        res = pipe(prompt_embeds=inp["prompt_embeds"][sl], negative_prompt_embeds=inp["negative_prompt_embeds"][sl],
                   image=inp["image"][sl], mask=inp["mask"][sl], depth=inp["depth"][sl], num_inference_steps=50,
                   guidance_scale=7.5, latents=inp["latents"][sl].clone(), output_type="pt", brushnet_conditioning_scale=1.0,
                   callback_on_step_end=cb, height=512, width=512)
    finally:
        ref_vae.randn_tensor = orig
    C3 = np.load(os.path.join(GOLD, "sd15_config1_slice.npz"))
    print("[config1 50] first three steps vs the 3-step fixture of the same inputs (different timestep grid, so only the "
          "shapes agree):", [tuple(trace[i].shape) == tuple(C3[f"latents_{i}"].shape) for i in range(3)])
    out = dict(timesteps=pipe.scheduler.timesteps.numpy(), steps=np.array(C1_50_STEPS))
    for n in C1_50_STEPS:
        out[f"latents_{n}"] = trace[n - 1].numpy()
    s_ = summarize(res.images, 4096)
    out["image_sample"], out["image_stats"] = s_["sample"], np.array([s_["sum"], s_["abssum"], s_["sample_stride"]])
    np.savez_compressed(os.path.join(GOLD, "sd15_config1_50steps.npz"), **out)
    print("configs[1] 50-step fixture written; final |latents| max", float(trace[-1].abs().max()))


class _StopAfter(Exception):
    pass


def _sdxl_full_models(dtype=None):
    """Full-size SDXL-base UNet + BrushNet-XL (5 conditioning channels: masked-image latents + mask,
    pipeline_brushnet_sd_xl.py:1301-1310) + SDXL VAE with the seeded synthetic weights (seeds 30 / 31 / 32)."""
    ucfg, vcfg = R.SDXL_UNET, R.SDXL_VAE
    unet = build_unet(ucfg)
    load_synth(unet, 30)
    brushnet = BrushNetModel.from_unet(unet, conditioning_channels=5, load_weights_from_unet=False).eval()
    load_synth(brushnet, 31)
    vae = build_vae(vcfg)
    load_synth(vae, 32)
    if dtype is not None:
        for m in (unet, brushnet, vae):
            m.to(dtype)
    return unet, brushnet, vae


def sdxl_inputs():
    inp = synth.pipeline_inputs(2, 1024, 1024, seed=4242, cross_dim=2048)
    gp = torch.Generator().manual_seed(4243)
    inp["pooled"], inp["npooled"] = torch.randn(2, 1280, generator=gp), torch.randn(2, 1280, generator=gp)
    return inp


def run_sdxl_full(pipe_cls, unet, brushnet, vae, cast=None, steps_kept=2):
    """BASELINE.json configs[4] sizes for ONE image: image 0 of the batch-2 x 1024 x 1024 synthetic inputs through the
    reference's StableDiffusionXLBrushNetPipeline (pipeline_brushnet_sd_xl.py:936-1535) on the 30-step DDIM grid, CFG 5.0;
    the run is aborted after `steps_kept` steps (each costs ~16 TFLOP on the CPU).  Returns the latents after each kept step."""
    import diffusers.models.autoencoders.vae as ref_vae
    sched = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                          clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = pipe_cls(vae=vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, unet=unet,
                    brushnet=brushnet, scheduler=sched, force_zeros_for_empty_prompt=True, add_watermarker=False)
    pipe.set_progress_bar_config(disable=True)
    inp = sdxl_inputs()
    sl = slice(0, 1)
    nz = inp["vae_noise"]
    noise = torch.cat([nz[:2][sl], nz[2:][sl]])
    c = (lambda t: t.to(cast)) if cast is not None else (lambda t: t)
    orig = ref_vae.randn_tensor
    ref_vae.randn_tensor = lambda shape, generator=None, device=None, dtype=None, layout=None: noise.to(dtype)
    trace = []
    import time
    t0 = time.time()

    def cb(p_, i, t_, kw_):
        trace.append(kw_["latents"].float().clone())
        print(f"[sdxl full] step {i + 1} at {time.time() - t0:.0f} s", flush=True)
        if len(trace) >= steps_kept:
            raise _StopAfter()
        return {}

    try:
        pipe(prompt_embeds=c(inp["prompt_embeds"][sl]), negative_prompt_embeds=c(inp["negative_prompt_embeds"][sl]),
             pooled_prompt_embeds=c(inp["pooled"][sl]), negative_pooled_prompt_embeds=c(inp["npooled"][sl]), image=inp["image"][sl],
             mask=inp["mask"][sl], num_inference_steps=30, guidance_scale=5.0, latents=c(inp["latents"][sl].clone()),
             output_type="latent", brushnet_conditioning_scale=1.0, height=1024, width=1024, callback_on_step_end=cb)
    except _StopAfter:
        pass
    finally:
        ref_vae.randn_tensor = orig
    return trace, pipe.scheduler.timesteps


def sdxl_full():
    from diffusers.pipelines.brushnet.pipeline_brushnet_sd_xl import StableDiffusionXLBrushNetPipeline
    unet, brushnet, vae = _sdxl_full_models()
    with open(os.path.join(GOLD, "keys_sdxl.json"), "w") as f:
        json.dump(dict(unet={k: tuple(v.shape) for k, v in unet.state_dict().items()},
                       brushnet={k: tuple(v.shape) for k, v in brushnet.state_dict().items()},
                       vae={k: tuple(v.shape) for k, v in vae.state_dict().items()}), f, indent=0, sort_keys=True)
    trace, ts = run_sdxl_full(StableDiffusionXLBrushNetPipeline, unet, brushnet, vae)
    out = dict(timesteps=ts.numpy())
    for i, l in enumerate(trace):
        out[f"latents_{i}"] = l.numpy()
    np.savez_compressed(os.path.join(GOLD, "sdxl_config4_slice.npz"), **out)
    print("configs[4] (SDXL full width) fixture written; |latents| max", [float(l.abs().max()) for l in trace])


def frontend_inputs():
    """Seeded inputs of the image front-end fixture (regenerated by the tests; only the PIL source is stored, as uint8)."""
    g = torch.Generator().manual_seed(515)
    t01 = torch.rand(2, 3, 40, 56, generator=g)                        # tensor in [0, 1]
    tneg = torch.rand(2, 3, 40, 56, generator=g) * 2.0 - 1.0            # tensor that already holds negatives
    mask = (torch.rand(2, 3, 40, 56, generator=g) > 0.6).float()        # a mask-like tensor
    arr = torch.rand(40, 56, 3, generator=g).numpy()                    # one HWC float image
    u8 = (torch.rand(40, 56, 3, generator=g) * 255).to(torch.uint8).numpy()     # the PIL image's pixels
    post = torch.rand(2, 3, 24, 32, generator=g) * 2.4 - 1.2            # decoder output incl. values outside [-1, 1]
    return dict(t01=t01, tneg=tneg, mask=mask, arr=arr, u8=u8, post=post)


def frontend():
    """The reference's REAL VaeImageProcessor (image_processor.py:446-610: preprocess with its resize / normalise rules for
    torch / numpy / PIL inputs, postprocess to pt / np / pil) on seeded inputs -> tests/golden/frontend.npz.  The device
    front-end kernels (csrc/frontend.hip) and the host class of the package are checked against THESE outputs."""
    import PIL.Image
    from diffusers.image_processor import VaeImageProcessor
    vp = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True)
    I = frontend_inputs()
    out = dict(pil_u8=I["u8"])
    for name in ("t01", "tneg", "mask"):
        out[f"{name}_same"] = vp.preprocess(I[name], height=40, width=56).numpy()          # no resize
        out[f"{name}_resized"] = vp.preprocess(I[name], height=32, width=48).numpy()       # F.interpolate nearest (:resize)
        out[f"{name}_default"] = vp.preprocess(I[name]).numpy()                            # 40 x 56: both multiples of 8
    out["np_resized"] = vp.preprocess(I["arr"], height=32, width=48).numpy()
    out["np_list"] = vp.preprocess([I["arr"], I["arr"][::-1].copy()], height=40, width=56).numpy()
    pil = PIL.Image.fromarray(I["u8"])
    out["pil_same"] = vp.preprocess(pil, height=40, width=56).numpy()
    out["pil_resized"] = vp.preprocess(pil, height=32, width=48).numpy()                   # PIL lanczos
    out["post_pt"] = vp.postprocess(I["post"], output_type="pt", do_denormalize=[True, True]).numpy()
    out["post_np"] = vp.postprocess(I["post"], output_type="np", do_denormalize=[True, True])
    out["post_pil"] = np.stack([np.array(im) for im in vp.postprocess(I["post"], output_type="pil", do_denormalize=[True, True])])
    out["post_pt_mixed"] = vp.postprocess(I["post"], output_type="pt", do_denormalize=[True, False]).numpy()
    np.savez_compressed(os.path.join(GOLD, "frontend.npz"), **out)
    print("front-end fixture written:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-config1", action="store_true")
    ap.add_argument("--only-config1-50", action="store_true", help="one image of configs[1] through all 50 DDIM steps (~10-45 min of CPU)")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only-full", action="store_true")
    ap.add_argument("--only-xl", action="store_true")
    ap.add_argument("--only-frontend", action="store_true")
    ap.add_argument("--only-sdxl-full", action="store_true", help="one image of configs[4] (SDXL 1024^2) for 2 steps, full width (~5 min, ~25 GB)")
    ap.add_argument("--only-train", action="store_true")
    ap.add_argument("--only-layers", action="store_true")
    ap.add_argument("--only-guess", action="store_true")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if a.only_config1:
        config1_slice()
        sys.exit(0)
    if a.only_config1_50:
        config1_50steps()
        sys.exit(0)
    if a.only_xl:
        tiny_xl()
        sys.exit(0)
    if a.only_frontend:
        frontend()
        sys.exit(0)
    if a.only_sdxl_full:
        sdxl_full()
        sys.exit(0)
    if a.only_train:
        tiny_train()
        sys.exit(0)
    if a.only_layers:
        layers_full()
        sys.exit(0)
    if a.only_guess:
        tiny_guess_mode()
        sys.exit(0)
    if not a.only_full:
        tiny()
        frontend()
        tiny_guess_mode()
        tiny_train()
        layers_full()
        tiny_xl()
    if a.full or a.only_full:
        full()
        config1_slice()
