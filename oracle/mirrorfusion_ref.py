"""CPU ORACLE — test infrastructure, NOT product code.

A plain PyTorch-CPU fp32 restatement of the reference's MirrorFusion hot path (SD1.5 UNet + BrushNet
dual-branch denoising, DDIM / PNDM step, AutoencoderKL encode / decode), written against flat
state dicts that use the reference's own key names.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; the product (`reflecting-reality_amd/`) never
does, and it fails loudly when its HIP extension is missing rather than fall back to anything here.

Pinning: every function is checked against outputs of the *imported reference itself*
(`tools/make_golden.py`, run in the build container where /root/reference exists) stored as fixtures
under `tests/golden/` — see tests/test_oracle_golden.py — and the scheduler restatements are also
checked against the reference's own known-answer tests (tests/schedulers/test_scheduler_ddim.py:122-153,
test_scheduler_pndm.py:210-242).  All `file:line` citations are relative to
/root/reference/MirrorFusion/src/diffusers/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ---------------------------------------------------------------------------------------------
# configs (the subset of config.json fields the hot path reads)
# ---------------------------------------------------------------------------------------------
SD15_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0)
SD15_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
SD15_SCHED = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                  steps_offset=1, set_alpha_to_one=False, clip_sample=False, skip_prk_steps=True)

TINY_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
    cross_attention_dim=32, attention_head_dim=4, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0)
TINY_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64), layers_per_block=1,
                norm_num_groups=32, scaling_factor=0.18215)
# SDXL (stabilityai/stable-diffusion-xl-base-1.0 unet/config.json) and a tiny configuration of the same architecture:
# linear proj_in/proj_out, per-level transformer depth and head count, text_time additional embedding
SDXL_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    cross_attention_dim=2048, attention_head_dim=(5, 10, 20), transformer_layers_per_block=(1, 2, 10),
    use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=256,
    projection_class_embeddings_input_dim=2816, norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0)
SDXL_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.13025)     # stabilityai/stable-diffusion-xl-base-1.0 vae/config.json
TINY_XL_UNET = dict(
    in_channels=4, out_channels=4, block_out_channels=(32, 64, 64), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    cross_attention_dim=48, attention_head_dim=(2, 4, 8), transformer_layers_per_block=(1, 2, 3),
    use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=8,
    projection_class_embeddings_input_dim=6 * 8 + 24, norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0)


def brushnet_config(unet_cfg: dict, conditioning_channels: int = 6) -> dict:
    """BrushNetModel.from_unet (models/brushnet.py:452-530): same widths, attention-free blocks."""
    n = len(unet_cfg["block_out_channels"])
    cfg = dict(unet_cfg)
    cfg.update(conditioning_channels=conditioning_channels, down_block_types=("DownBlock2D",) * n,
               up_block_types=("UpBlock2D",) * n, mid_block_type="MidBlock2D")
    return cfg


def _heads(cfg: dict, level: int) -> int:
    h = cfg["attention_head_dim"]          # (sic) SD1.5 stores the head COUNT here (unet_2d_condition.py:~300)
    return h if isinstance(h, int) else h[level]


# ---------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float) -> torch.Tensor:
    """models/embeddings.py:27-67 (scale 1, max_period 10000)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def time_embed(sd: SD, cfg: dict, timestep, batch: int, added: Optional[dict] = None) -> torch.Tensor:
    """Timesteps + TimestepEmbedding (embeddings.py:191-254; brushnet.py:750-772, unet_2d_condition.py:1154), plus the
    SDXL 'text_time' additional embedding (unet_2d_condition.py:971-987, brushnet.py:789-805): Fourier features of the six
    time ids concatenated to the pooled text embedding, through add_embedding, added to the time embedding."""
    t = torch.as_tensor(timestep)
    if t.dim() == 0:
        t = t[None]
    t = t.expand(batch)
    c0 = cfg["block_out_channels"][0]
    e = timestep_embedding(t, c0, cfg["flip_sin_to_cos"], cfg["freq_shift"])
    e = F.linear(e, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    e = F.silu(e)
    e = F.linear(e, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    if cfg.get("addition_embed_type") == "text_time":
        text_embeds, time_ids = added["text_embeds"], added["time_ids"]
        te = timestep_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        a = torch.cat([text_embeds, te.reshape(text_embeds.shape[0], -1)], -1)
        a = F.silu(F.linear(a, sd["add_embedding.linear_1.weight"], sd["add_embedding.linear_1.bias"]))
        e = e + F.linear(a, sd["add_embedding.linear_2.weight"], sd["add_embedding.linear_2.bias"])
    return e


def resnet(sd: SD, p: str, x: torch.Tensor, temb: Optional[torch.Tensor], groups: int, eps: float) -> torch.Tensor:
    """ResnetBlock2D.forward, default time-embedding norm, output_scale_factor 1 (models/resnet.py:329-405)."""
    h = F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    if temb is not None:
        t = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
        h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.silu(h)
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def attention(sd: SD, p: str, x: torch.Tensor, ctx: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """Attention + AttnProcessor2_0 on [B, S, C] (models/attention_processor.py:1213-1286)."""
    ctx = x if ctx is None else ctx
    q = F.linear(x, sd[p + "to_q.weight"], sd.get(p + "to_q.bias"))
    k = F.linear(ctx, sd[p + "to_k.weight"], sd.get(p + "to_k.bias"))
    v = F.linear(ctx, sd[p + "to_v.weight"], sd.get(p + "to_v.bias"))
    b, s, c = q.shape
    d = c // heads
    q = q.view(b, s, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(b, s, c)
    return F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def basic_transformer_block(sd: SD, p: str, x: torch.Tensor, ehs: torch.Tensor, heads: int) -> torch.Tensor:
    """BasicTransformerBlock, layer_norm variant, GEGLU feed-forward (models/attention.py:291-412,617-675)."""
    c = x.shape[-1]
    n = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    x = attention(sd, p + "attn1.", n, None, heads) + x
    n = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    x = attention(sd, p + "attn2.", n, ehs, heads) + x
    n = F.layer_norm(x, (c,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    hg = F.linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    h, gate = hg.chunk(2, dim=-1)                      # models/activations.py:100-103
    ff = F.linear(h * F.gelu(gate), sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"])
    return ff + x


def transformer_2d(sd: SD, p: str, x: torch.Tensor, ehs: torch.Tensor, heads: int, groups: int) -> torch.Tensor:
    """Transformer2DModel, continuous input, conv projections (models/transformers/transformer_2d.py:334-430)."""
    b, c, hh, ww = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    linear = sd[p + "proj_in.weight"].dim() == 2                   # use_linear_projection (transformer_2d.py:376-385)
    if linear:
        h = F.linear(h.permute(0, 2, 3, 1).reshape(b, hh * ww, c), sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    else:
        h = F.conv2d(h, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    i = 0
    while f"{p}transformer_blocks.{i}.norm1.weight" in sd:
        h = basic_transformer_block(sd, f"{p}transformer_blocks.{i}.", h, ehs, heads)
        i += 1
    if linear:                                                     # transformer_2d.py:418-427
        h = F.linear(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
    else:
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        h = F.conv2d(h, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return h + res


def downsample(sd: SD, p: str, x: torch.Tensor, padding: int = 1) -> torch.Tensor:
    """Downsample2D with conv (models/downsampling.py:134-154); padding 0 => asymmetric (0,1,0,1) pad."""
    if padding == 0:
        x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=padding)


def upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Upsample2D: nearest 2x then conv3x3 (models/upsampling.py:143-193)."""
    x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)


# ---------------------------------------------------------------------------------------------
# BrushNet (models/brushnet.py:678-925)
# ---------------------------------------------------------------------------------------------
def brushnet_forward(sd: SD, cfg: dict, sample: torch.Tensor, timestep, brushnet_cond: torch.Tensor,
                     conditioning_scale: float = 1.0, added: Optional[dict] = None, guess_mode: bool = False
                     ) -> Tuple[List[torch.Tensor], torch.Tensor, List[torch.Tensor]]:
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    nlev = len(cfg["block_out_channels"])
    lpb = cfg["layers_per_block"]
    emb = time_embed(sd, cfg, timestep, sample.shape[0], added)
    x = torch.cat([sample, brushnet_cond], 1)                                          # :810
    x = F.conv2d(x, sd["conv_in_condition.weight"], sd["conv_in_condition.bias"], padding=1)
    down_res = [x]
    for i in range(nlev):                                                              # :815-828 (DownBlock2D)
        for j in range(lpb):
            x = resnet(sd, f"down_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            down_res.append(x)
        if i != nlev - 1:
            x = downsample(sd, f"down_blocks.{i}.downsamplers.0.", x)
            down_res.append(x)
    bn_down = [F.conv2d(r, sd[f"brushnet_down_blocks.{k}.weight"], sd[f"brushnet_down_blocks.{k}.bias"])
               for k, r in enumerate(down_res)]                                        # :831-834
    for j in range(2):                                                                 # MidBlock2D, unet_2d_blocks.py:1082-1111
        x = resnet(sd, f"mid_block.resnets.{j}.", x, emb, g, eps)
    bn_mid = F.conv2d(x, sd["brushnet_mid_block.weight"], sd["brushnet_mid_block.bias"])   # :851
    up_res: List[torch.Tensor] = []
    skips = list(down_res)
    for i in range(nlev):                                                              # :856-887 (UpBlock2D, return_res_samples)
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], 1)
            x = resnet(sd, f"up_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            up_res.append(x)
        if i != nlev - 1:
            x = upsample(sd, f"up_blocks.{i}.upsamplers.0.", x)
            up_res.append(x)
    bn_up = [F.conv2d(r, sd[f"brushnet_up_blocks.{k}.weight"], sd[f"brushnet_up_blocks.{k}.bias"])
             for k, r in enumerate(up_res)]                                            # :890-893
    if guess_mode:                                                                     # :896-902: 0.1 ... 1.0, log-spaced
        scales = torch.logspace(-1, 0, len(bn_down) + 1 + len(bn_up)) * conditioning_scale
        nd = len(bn_down)
        return ([d * sc for d, sc in zip(bn_down, scales[:nd])], bn_mid * scales[nd],
                [u * sc for u, sc in zip(bn_up, scales[nd + 1:])])
    s = conditioning_scale                                                             # :904-906
    return [d * s for d in bn_down], bn_mid * s, [u * s for u in bn_up]


# ---------------------------------------------------------------------------------------------
# UNet with BrushNet injection (models/unets/unet_2d_condition.py:1039-1348)
# ---------------------------------------------------------------------------------------------
def unet_forward(sd: SD, cfg: dict, sample: torch.Tensor, timestep, ehs: torch.Tensor,
                 down_add: Optional[Sequence[torch.Tensor]] = None, mid_add: Optional[torch.Tensor] = None,
                 up_add: Optional[Sequence[torch.Tensor]] = None, added: Optional[dict] = None) -> torch.Tensor:
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    nlev = len(cfg["block_out_channels"])
    lpb = cfg["layers_per_block"]
    is_brushnet = down_add is not None and mid_add is not None and up_add is not None      # :1202
    down_add = list(down_add) if is_brushnet else None
    up_add = list(up_add) if is_brushnet else None
    emb = time_embed(sd, cfg, timestep, sample.shape[0], added)
    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [x]                                                                            # :1215 (captured pre-add)
    if is_brushnet:
        x = x + down_add.pop(0)                                                            # :1218
    for i, btype in enumerate(cfg["down_block_types"]):
        has_attn = btype == "CrossAttnDownBlock2D"
        for j in range(lpb):
            x = resnet(sd, f"down_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            if has_attn:
                x = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}.", x, ehs, _heads(cfg, i), g)
            if is_brushnet:
                x = x + down_add.pop(0)                                                    # unet_2d_blocks.py:1388-1389,1483-1484
            skips.append(x)
        if i != nlev - 1:
            x = downsample(sd, f"down_blocks.{i}.downsamplers.0.", x)
            if is_brushnet:
                x = x + down_add.pop(0)                                                    # :1397-1398,1492-1493
            skips.append(x)
    # UNetMidBlock2DCrossAttn (unet_2d_blocks.py:850-899)
    x = resnet(sd, "mid_block.resnets.0.", x, emb, g, eps)
    x = transformer_2d(sd, "mid_block.attentions.0.", x, ehs, _heads(cfg, nlev - 1), g)
    x = resnet(sd, "mid_block.resnets.1.", x, emb, g, eps)
    if is_brushnet:
        x = x + mid_add                                                                    # :1288-1289
    for i, btype in enumerate(cfg["up_block_types"]):
        has_attn = btype == "CrossAttnUpBlock2D"
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], 1)                                             # unet_2d_blocks.py:2586,2728
            x = resnet(sd, f"up_blocks.{i}.resnets.{j}.", x, emb, g, eps)
            if has_attn:
                x = transformer_2d(sd, f"up_blocks.{i}.attentions.{j}.", x, ehs, _heads(cfg, nlev - 1 - i), g)
            if is_brushnet:
                x = x + up_add.pop(0)                                                      # :2626-2627,2751-2752
        if i != nlev - 1:
            x = upsample(sd, f"up_blocks.{i}.upsamplers.0.", x)
            if is_brushnet:
                x = x + up_add.pop(0)                                                      # :2634-2635,2760-2761
    x = F.group_norm(x, g, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)      # :1336-1339
    x = F.silu(x)
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


# ---------------------------------------------------------------------------------------------
# AutoencoderKL (models/autoencoders/autoencoder_kl.py:238-309, vae.py:46-348)
# ---------------------------------------------------------------------------------------------
def _vae_mid(sd: SD, p: str, x: torch.Tensor, groups: int) -> torch.Tensor:
    """UNetMidBlock2D: resnet, single-head spatial attention with group norm + residual, resnet
    (unet_2d_blocks.py:601-753; attention_processor.py:1228-1284 with input_ndim == 4)."""
    x = resnet(sd, p + "resnets.0.", x, None, groups, 1e-6)
    b, c, hh, ww = x.shape
    a = p + "attentions.0."
    res = x
    h = x.view(b, c, hh * ww).transpose(1, 2)
    h = F.group_norm(h.transpose(1, 2), groups, sd[a + "group_norm.weight"], sd[a + "group_norm.bias"], 1e-6).transpose(1, 2)
    h = attention(sd, a, h, None, 1)
    h = h.transpose(-1, -2).reshape(b, c, hh, ww)
    x = h + res
    return resnet(sd, p + "resnets.1.", x, None, groups, 1e-6)


def vae_encode_moments(sd: SD, cfg: dict, x: torch.Tensor) -> torch.Tensor:
    """Encoder.forward + quant_conv -> moments [B, 2*latent, h, w] (vae.py:140-182, autoencoder_kl.py:238-268)."""
    g = cfg["norm_num_groups"]
    nlev = len(cfg["block_out_channels"])
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(nlev):
        for j in range(cfg["layers_per_block"]):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", h, None, g, 1e-6)
        if i != nlev - 1:
            h = downsample(sd, f"encoder.down_blocks.{i}.downsamplers.0.", h, padding=0)
    h = _vae_mid(sd, "encoder.mid_block.", h, g)
    h = F.group_norm(h, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6)
    h = F.silu(h)
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def vae_sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample with explicit noise (vae.py:769-791)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def vae_decode(sd: SD, cfg: dict, z: torch.Tensor) -> torch.Tensor:
    """post_quant_conv + Decoder.forward (autoencoder_kl.py:270-309, vae.py:285-348)."""
    g = cfg["norm_num_groups"]
    nlev = len(cfg["block_out_channels"])
    h = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _vae_mid(sd, "decoder.mid_block.", h, g)
    for i in range(nlev):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", h, None, g, 1e-6)
        if i != nlev - 1:
            h = upsample(sd, f"decoder.up_blocks.{i}.upsamplers.0.", h)
    h = F.group_norm(h, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6)
    h = F.silu(h)
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


# ---------------------------------------------------------------------------------------------
# schedulers
# ---------------------------------------------------------------------------------------------
def _alphas_cumprod(cfg: dict) -> torch.Tensor:
    n = cfg.get("num_train_timesteps", 1000)
    if cfg.get("beta_schedule", "linear") == "scaled_linear":
        betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
    else:
        betas = torch.linspace(cfg.get("beta_start", 0.0001), cfg.get("beta_end", 0.02), n, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


class DDIMRef:
    """schedulers/scheduling_ddim.py: set_timesteps ('leading') :299-342, step (eta 0) :344-470."""

    def __init__(self, **cfg):
        self.cfg = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                        clip_sample=True, set_alpha_to_one=True, steps_offset=0, clip_sample_range=1.0,
                        prediction_type="epsilon")
        self.cfg.update({k: v for k, v in cfg.items() if k in self.cfg})
        self.alphas_cumprod = _alphas_cumprod(self.cfg)
        self.final_alpha_cumprod = torch.tensor(1.0) if self.cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        ratio = self.cfg["num_train_timesteps"] // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.cfg["steps_offset"]
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
        timestep = int(timestep)
        prev_t = timestep - self.cfg["num_train_timesteps"] // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.cfg["prediction_type"] == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        else:  # v_prediction (:418-420)
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
            eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
        if self.cfg["clip_sample"]:
            x0 = x0.clamp(-self.cfg["clip_sample_range"], self.cfg["clip_sample_range"])
        direction = (1 - a_prev) ** 0.5 * eps                   # eta = 0 -> std_dev_t = 0
        return a_prev ** 0.5 * x0 + direction


class PNDMRef:
    """schedulers/scheduling_pndm.py: set_timesteps :168-226, step_prk :261-319, step_plms :321-390,
    _get_prev_sample :407-448."""

    def __init__(self, **cfg):
        self.cfg = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                        skip_prk_steps=False, set_alpha_to_one=False, steps_offset=0, prediction_type="epsilon")
        self.cfg.update({k: v for k, v in cfg.items() if k in self.cfg})
        self.alphas_cumprod = _alphas_cumprod(self.cfg)
        self.final_alpha_cumprod = torch.tensor(1.0) if self.cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4

    def set_timesteps(self, n: int):
        self.num_inference_steps = n
        nt = self.cfg["num_train_timesteps"]
        _ts = (np.arange(0, n) * (nt // n)).round() + self.cfg["steps_offset"]
        if self.cfg["skip_prk_steps"]:
            self.prk_timesteps = np.array([])
            self.plms_timesteps = np.concatenate([_ts[:-1], _ts[-2:-1], _ts[-1:]])[::-1].copy()
        else:
            prk = np.array(_ts[-self.pndm_order:]).repeat(2) + np.tile(np.array([0, nt // n // 2]), self.pndm_order)
            self.prk_timesteps = (prk[:-1].repeat(2)[1:-1])[::-1].copy()
            self.plms_timesteps = _ts[:-3][::-1].copy()
        self.timesteps = torch.from_numpy(np.concatenate([self.prk_timesteps, self.plms_timesteps]).astype(np.int64))
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_model_output = 0
        self.cur_sample = None

    def _prev(self, sample, t, prev_t, eps):
        t, prev_t = int(t), int(prev_t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if self.cfg["prediction_type"] == "v_prediction":
            eps = (a_t ** 0.5) * eps + (b_t ** 0.5) * sample
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff * sample - (a_prev - a_t) * eps / denom

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor) -> torch.Tensor:
        if self.counter < len(self.prk_timesteps) and not self.cfg["skip_prk_steps"]:
            return self.step_prk(model_output, timestep, sample)
        return self.step_plms(model_output, timestep, sample)

    def step_prk(self, model_output, timestep, sample):
        timestep = int(timestep)
        ratio = self.cfg["num_train_timesteps"] // self.num_inference_steps
        diff_to_prev = 0 if self.counter % 2 else ratio // 2
        prev_t = timestep - diff_to_prev
        timestep = int(self.prk_timesteps[self.counter // 4 * 4])
        if self.counter % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 6 * model_output
            self.ets.append(model_output)
            self.cur_sample = sample
        elif (self.counter - 1) % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
        elif (self.counter - 2) % 4 == 0:
            self.cur_model_output = self.cur_model_output + 1 / 3 * model_output
        elif (self.counter - 3) % 4 == 0:
            model_output = self.cur_model_output + 1 / 6 * model_output
            self.cur_model_output = 0
        cur_sample = self.cur_sample if self.cur_sample is not None else sample
        out = self._prev(cur_sample, timestep, prev_t, model_output)
        self.counter += 1
        return out

    def step_plms(self, model_output, timestep, sample):
        timestep = int(timestep)
        ratio = self.cfg["num_train_timesteps"] // self.num_inference_steps
        prev_t = timestep - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = timestep
            timestep = timestep + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        out = self._prev(sample, timestep, prev_t, model_output)
        self.counter += 1
        return out


class UniPCRef:
    """schedulers/scheduling_unipc_multistep.py (bh1/bh2, predict_x0, lower_order_final, no Karras sigmas):
    set_timesteps :229-300, convert_model_output :385-453, predictor :455-582, corrector :584-719, step :754-830."""

    def __init__(self, **cfg):
        self.cfg = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                        solver_order=2, prediction_type="epsilon", predict_x0=True, solver_type="bh2",
                        lower_order_final=True, timestep_spacing="linspace", steps_offset=0)
        self.cfg.update({k: v for k, v in cfg.items() if k in self.cfg})
        self.alphas_cumprod = _alphas_cumprod(self.cfg)
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        nt = self.cfg["num_train_timesteps"]
        if self.cfg["timestep_spacing"] == "linspace":
            ts = np.linspace(0, nt - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        else:  # leading
            ts = (np.arange(0, n + 1) * (nt // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.cfg["steps_offset"]
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        last = ((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5
        self.sigmas = torch.from_numpy(np.concatenate([sigmas, [last]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * self.cfg["solver_order"]
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _convert(self, model_output, sample):
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[self.step_index])
        if self.cfg["prediction_type"] == "epsilon":
            return (sample - sigma_t * model_output) / alpha_t
        return alpha_t * sample - sigma_t * model_output          # v_prediction

    def _rb(self, order, idx_t, idx_s0, hist_offset):
        """Shared coefficient set-up of the predictor (:507-548) and corrector (:642-690)."""
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[idx_t])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[idx_s0])
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks = []
        for i in range(1, order):
            a_si, s_si = self._alpha_sigma(self.sigmas[self.step_index - (i + hist_offset)])
            rks.append((torch.log(a_si) - torch.log(s_si) - lambda_s0) / h)
        rks_t = torch.tensor([float(r) for r in rks] + [1.0])
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.cfg["solver_type"] == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks_t, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return dict(alpha_t=alpha_t, sigma_t=sigma_t, sigma_s0=sigma_s0, h_phi_1=h_phi_1, B_h=B_h, rks=rks,
                    R=torch.stack(R), b=torch.tensor([float(v) for v in b]))

    def step(self, model_output, timestep, sample):
        order_cfg = self.cfg["solver_order"]
        use_corrector = self.step_index > 0 and self.last_sample is not None
        m_t = self._convert(model_output, sample)
        if use_corrector:
            order = self.this_order
            c = self._rb(order, self.step_index, self.step_index - 1, 1)
            m0 = self.model_outputs[-1]
            x = self.last_sample
            D1s = [(self.model_outputs[-(i + 1)] - m0) / c["rks"][i - 1] for i in range(1, order)]
            rhos = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(c["R"], c["b"])
            x_t_ = c["sigma_t"] / c["sigma_s0"] * x - c["alpha_t"] * c["h_phi_1"] * m0
            corr = sum(rhos[i] * D1s[i] for i in range(len(D1s))) if D1s else 0
            sample = x_t_ - c["alpha_t"] * c["B_h"] * (corr + rhos[-1] * (m_t - m0))
        for i in range(order_cfg - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = m_t
        this_order = min(order_cfg, len(self.timesteps) - self.step_index) if self.cfg["lower_order_final"] else order_cfg
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        order = self.this_order
        c = self._rb(order, self.step_index + 1, self.step_index, 0)
        m0 = self.model_outputs[-1]
        D1s = [(self.model_outputs[-(i + 1)] - m0) / c["rks"][i - 1] for i in range(1, order)]
        x_t = c["sigma_t"] / c["sigma_s0"] * sample - c["alpha_t"] * c["h_phi_1"] * m0
        if D1s:
            rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(c["R"][:-1, :-1], c["b"][:-1])
            x_t = x_t - c["alpha_t"] * c["B_h"] * sum(rhos_p[i] * D1s[i] for i in range(len(D1s)))
        if self.lower_order_nums < order_cfg:
            self.lower_order_nums += 1
        self.step_index += 1
        return x_t


# ---------------------------------------------------------------------------------------------
# pipeline (pipelines/brushnet/pipeline_brushnet.py:848-1363, tensor inputs, prompt_embeds given)
# ---------------------------------------------------------------------------------------------
def preprocess_image(img: torch.Tensor) -> torch.Tensor:
    """VaeImageProcessor.preprocess for a [B,C,H,W] tensor (image_processor.py:532-555): values in [0,1] are
    mapped to [-1,1]; a tensor that already has negatives passes through (do_normalize skipped with a warning)."""
    if img.min() < 0:
        return img
    return 2.0 * img - 1.0


def build_conditioning(vae_sd: SD, vae_cfg: dict, image: torch.Tensor, mask: torch.Tensor,
                       depth: Optional[torch.Tensor], vae_noise: torch.Tensor, cfg_dup: bool = True,
                       depth_mode: str = "concat", depth_noise: Optional[torch.Tensor] = None,
                       normals: Optional[torch.Tensor] = None, normals_mode: Optional[str] = None,
                       normals_noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pipeline_brushnet.py:1116-1215.

    image/mask: [B,3,H,W] in [0,1]; depth [B,1,H,W] in [-1,1]; normals [B,3,H,W]; *_noise: posterior noise for the
    CFG-duplicated batch [2B,4,h,w] (the reference draws them from the global RNG in the order image (:1188),
    depth (:1206), normals (:1214)).  depth_mode "concat": nearest-resized depth (+1 ch, :1198-1202); "latents":
    depth repeated to 3 channels and VAE-encoded (+4 ch, :1203-1206); normals alike (:1208-1215, +3 / +4 ch).
    Returns conditioning_latents [2B, 4+1+..., h, w].
    """
    img = preprocess_image(image.float())
    m = preprocess_image(mask.float())
    if cfg_dup:
        img, m = torch.cat([img] * 2), torch.cat([m] * 2)
    original_mask = (m.sum(1)[:, None, :, :] < 0).to(img.dtype)                          # :1139
    moments = vae_encode_moments(vae_sd, vae_cfg, img)
    cond = vae_sample(moments, vae_noise) * vae_cfg["scaling_factor"]                    # :1188
    mm = F.interpolate(original_mask, size=cond.shape[-2:])                              # :1189-1195
    cond = torch.cat([cond, mm], 1)
    hw = cond.shape[-2:]
    if depth is not None:
        d = preprocess_image(depth.float())
        if cfg_dup:
            d = torch.cat([d] * 2)
        if depth_mode == "concat":
            cond = torch.cat([cond, F.interpolate(d, size=hw)], 1)                       # :1198-1202
        else:
            dm = vae_encode_moments(vae_sd, vae_cfg, d.repeat(1, 3, 1, 1))               # :1204-1205
            cond = torch.cat([cond, vae_sample(dm, depth_noise) * vae_cfg["scaling_factor"]], 1)
    if normals is not None and normals_mode is not None:
        nrm = preprocess_image(normals.float())
        if cfg_dup:
            nrm = torch.cat([nrm] * 2)
        if normals_mode == "concat":
            cond = torch.cat([cond, F.interpolate(nrm, size=hw)], 1)                     # :1208-1212
        else:
            nm = vae_encode_moments(vae_sd, vae_cfg, nrm)                                # :1213-1215
            cond = torch.cat([cond, vae_sample(nm, normals_noise) * vae_cfg["scaling_factor"]], 1)
    return cond


def denoise(unet_sd: SD, unet_cfg: dict, bn_sd: SD, bn_cfg: dict, scheduler, latents: torch.Tensor,
            cond_latents: torch.Tensor, prompt_embeds_2b: torch.Tensor, num_steps: int, guidance_scale: float = 7.5,
            conditioning_scale: float = 1.0, trace: Optional[list] = None, added: Optional[dict] = None,
            guess_mode: bool = False) -> torch.Tensor:
    """The hot loop (pipeline_brushnet.py:1250-1332, pipeline_brushnet_sd_xl.py:1398-1500) with CFG;
    prompt_embeds_2b = cat([negative, positive]); `added` = SDXL's added_cond_kwargs for the duplicated batch.
    guess_mode (:1260-1264, 1287-1293): BrushNet sees only the conditional batch (cond_latents of the UN-duplicated
    images), its residuals are log-scaled and the unconditional half of the UNet gets zeros."""
    scheduler.set_timesteps(num_steps)
    latents = latents * scheduler.init_noise_sigma
    for t in scheduler.timesteps:
        x2 = torch.cat([latents] * 2)
        if guess_mode:
            down, mid, up = brushnet_forward(bn_sd, bn_cfg, latents, t, cond_latents, conditioning_scale, added, guess_mode=True)
            down = [torch.cat([torch.zeros_like(d), d]) for d in down]
            mid = torch.cat([torch.zeros_like(mid), mid])
            up = [torch.cat([torch.zeros_like(u), u]) for u in up]
        else:
            down, mid, up = brushnet_forward(bn_sd, bn_cfg, x2, t, cond_latents, conditioning_scale, added)
        eps = unet_forward(unet_sd, unet_cfg, x2, t, prompt_embeds_2b, down, mid, up, added)
        eu, ec = eps.chunk(2)
        eps = eu + guidance_scale * (ec - eu)                                            # :1310-1312
        latents = scheduler.step(eps, t, latents)                                        # :1315
        if trace is not None:
            trace.append(latents.clone())
    return latents


# ---------------------------------------------------------------------------------------------
# training step, forward half (examples/brushnet/train_brushnet_mirror.py)
# ---------------------------------------------------------------------------------------------
def training_loss(unet_sd: SD, unet_cfg: dict, bn_sd: SD, bn_cfg: dict, sched_cfg: dict, latents: torch.Tensor,
                  noise: torch.Tensor, timesteps: torch.Tensor, ehs: torch.Tensor, cond_latents: torch.Tensor,
                  snr_gamma: Optional[float] = None):
    """train_brushnet_mirror.py:1407-1449 (+ MirrorFusionModel.forward :858-888, DDPM add_noise
    scheduling_ddpm.py:501-525, get_velocity :527-546, compute_snr training_utils.py:50-73).
    Returns (loss, model_pred)."""
    ac = _alphas_cumprod(sched_cfg)
    sa = (ac[timesteps] ** 0.5).flatten()[:, None, None, None]
    sb = ((1 - ac[timesteps]) ** 0.5).flatten()[:, None, None, None]
    noisy = sa * latents + sb * noise                                                     # :1416
    down, mid, up = brushnet_forward(bn_sd, bn_cfg, noisy, timesteps, cond_latents, 1.0)  # :860-866
    pred = unet_forward(unet_sd, unet_cfg, noisy, timesteps, ehs, down, mid, up)          # :874-886
    ptype = sched_cfg.get("prediction_type", "epsilon")
    if ptype == "epsilon":
        target = noise
    elif ptype == "v_prediction":
        target = sa * noise - sb * latents
    else:
        raise ValueError(f"Unknown prediction type {ptype}")
    if snr_gamma is None:
        return F.mse_loss(pred.float(), target.float(), reduction="mean"), pred           # :1434
    alpha = (ac ** 0.5)[timesteps].float()
    sigma = ((1.0 - ac) ** 0.5)[timesteps].float()
    snr = (alpha / sigma) ** 2
    w = torch.stack([snr, snr_gamma * torch.ones_like(timesteps)], dim=1).min(dim=1)[0]   # :1441-1443
    w = w / snr if ptype == "epsilon" else w / (snr + 1)
    loss = F.mse_loss(pred.float(), target.float(), reduction="none")
    loss = loss.mean(dim=list(range(1, len(loss.shape)))) * w
    return loss.mean(), pred


# ---------------------------------------------------------------------------------------------
# training step, backward half: autograd through the restatement above + clip + AdamW
# (examples/brushnet/train_brushnet_mirror.py:1188-1200 optimizer, :1459-1466 backward / clip / step)
# ---------------------------------------------------------------------------------------------
def adamw_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """torch.optim.AdamW's single-tensor update (decoupled weight decay, bias correction, no amsgrad), in place."""
    b1, b2 = betas
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def training_steps(unet_sd: SD, unet_cfg: dict, bn_sd: SD, bn_cfg: dict, sched_cfg: dict, batches, lr: float = 1e-5,
                   max_grad_norm: float = 1.0, snr_gamma: Optional[float] = None, train_unet: bool = False,
                   betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """One optimizer step per batch (latents, noise, timesteps, ehs, cond): loss -> backward -> clip_grad_norm_(1.0) ->
    AdamW, on copies of the state dicts.  Returns (new_bn_sd, new_unet_sd, records) where records[i] holds the loss,
    the pre-clip gradient norm and the gradients of step i."""
    bn = {k: v.clone().requires_grad_(True) for k, v in bn_sd.items()}
    un = {k: v.clone().requires_grad_(train_unet) for k, v in unet_sd.items()}
    params = list(bn.items()) + ([("unet." + k, v) for k, v in un.items()] if train_unet else [])
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params}
    records = []
    for step, (latents, noise, timesteps, ehs, cond) in enumerate(batches, 1):
        for _, v in params:
            v.grad = None
        loss, _ = training_loss(un, unet_cfg, bn, bn_cfg, sched_cfg, latents, noise, timesteps, ehs, cond, snr_gamma)
        loss.backward()
        grads = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in params}
        total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()      # clip_grad_norm_ (:1463)
        coef = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
        with torch.no_grad():
            for k, v in params:
                adamw_update(v, grads[k] * coef, state[k][0], state[k][1], step, lr, betas, eps, weight_decay)
        records.append(dict(loss=float(loss.detach()), grad_norm=float(total), grads=grads))
    return ({k: v.detach() for k, v in bn.items()}, {k: v.detach() for k, v in un.items()}, records)


# ---------------------------------------------------------------------------------------------
# dataset-side depth transform (examples/brushnet/dataset/dataset.py:98-166)
# PARITY UNPINNED for this one function: the reference module imports h5py / torchvision / cv2, none of which exist in
# the build container, so it cannot be imported to check this restatement; it follows the numpy statements of
# :122-145 line by line (the torchvision Resize / CenterCrop of :150-164 are the identity at the native resolution).
# ---------------------------------------------------------------------------------------------
def _resize_center_crop_ref(t: torch.Tensor, resolution: int, antialias: bool = True) -> torch.Tensor:
    """transforms.Resize(resolution, BICUBIC) + CenterCrop(resolution) on a [C, H, W] tensor: smaller edge -> resolution,
    aspect kept (truncated), centre crop.  antialias=True is torchvision 0.18's default (the reference's pinned version):
    its tensor path calls exactly this interpolate (torchvision/transforms/_functional_tensor.py `resize`) at every scale;
    False is the plain kernel (torchvision < 0.17 for tensors)."""
    _, h, w = t.shape
    nh, nw = (resolution, int(resolution * w / h)) if h <= w else (int(resolution * h / w), resolution)
    if (nh, nw) != (h, w):
        t = F.interpolate(t[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=antialias)[0]
    top, left = int(round((nh - resolution) / 2.0)), int(round((nw - resolution) / 2.0))
    return t[:, top:top + resolution, left:left + resolution]


def apply_transforms_depth_ref(depth_map, mask=None, max_scene_depth: float = 5.0, norm_range=(-1, 1), delta: float = 0.5,
                               normalization_method: str = "max_scene_depth", resolution: Optional[int] = None,
                               antialias: bool = True):
    import numpy as np
    depth_map = np.copy(depth_map)
    if mask is not None and mask.ndim == 3:
        mask = mask[:, :, 0]                                                              # :111-112
    if normalization_method == "percentile":
        d_2, d_98 = np.percentile(depth_map, 2), np.percentile(depth_map, 98)             # :116-117
        clipped = np.clip(depth_map, d_2, d_98)                                           # :120
        if list(norm_range) == [0, 1]:
            out = (clipped - d_2) / (d_98 - d_2)                                          # :124
        elif list(norm_range) == [-1, 1]:
            out = 2.0 * (clipped - d_2) / (d_98 - d_2) - 1.0                              # :126
        else:
            raise ValueError("Unsupported normalization range. Use [0, 1] or [-1, 1].")
    else:
        if mask is not None:
            max_scene_depth = np.max(depth_map[mask > 0]) + delta                         # :129-134
        clipped = np.clip(depth_map, 0, max_scene_depth)                                  # :137
        if list(norm_range) == [0, 1]:
            out = clipped / max_scene_depth                                               # :141
        elif list(norm_range) == [-1, 1]:
            out = 2.0 * (clipped / max_scene_depth) - 1.0                                 # :143
        else:
            raise ValueError("Unsupported normalization range. Use [0, 1] or [-1, 1].")
    t = torch.tensor(out, dtype=torch.float32).unsqueeze(0)                               # :150
    return t if resolution is None else _resize_center_crop_ref(t, resolution, antialias)   # :152-164


def apply_transforms_normals_ref(normals_map, resolution: int = 512, antialias: bool = True):
    """dataset.py:184-192 (the map-valued modes): permute to CHW, Resize / CenterCrop, Normalize([0.5], [0.5]).  PARITY
    UNPINNED like apply_transforms_depth_ref (same module)."""
    t = torch.tensor(normals_map, dtype=torch.float32).permute(2, 0, 1)
    return (_resize_center_crop_ref(t, resolution, antialias) - 0.5) / 0.5
