"""Build-owned deterministic weights and inputs.

No pretrained checkpoint exists offline, so every run (golden generation against the imported
reference, parity tests on the GPU box, bench.py) draws parameters from this key-seeded CPU generator:
the value of a tensor depends only on (base seed, state-dict key, shape), never on creation order, so
the reference modules here and the HIP models on the GPU box see bit-identical fp32 parameters.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator(device="cpu").manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def fill(key: str, shape: Iterable[int], seed: int = 0) -> torch.Tensor:
    """Deterministic fp32 tensor for state-dict entry `key`."""
    shape = tuple(int(s) for s in shape)
    g = _gen(key, seed)
    parts = key.split(".")
    leaf = parts[-1]
    parent = parts[-2] if len(parts) >= 2 else ""
    is_norm = parent.startswith("norm") or parent in ("conv_norm_out", "group_norm")
    if leaf == "weight" and is_norm and len(shape) == 1:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return (0.1 if is_norm else 0.02) * torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        std = 1.0 / math.sqrt(fan_in)
        if key.startswith("brushnet_"):          # the reference zero-inits these; make the injection path live
            std *= 0.5
        return std * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)


def state_dict_for(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: fill(k, s, seed) for k, s in shapes.items()}


def pipeline_inputs(batch: int, height: int, width: int, seed: int = 1234, cross_dim: int = 768, seq: int = 77,
                    latent_channels: int = 4, vae_scale: int = 8) -> Dict[str, torch.Tensor]:
    """Synthetic pipeline inputs of SURVEY.md §8(d): all drawn on the CPU so they are device-independent."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    hl, wl = height // vae_scale, width // vae_scale
    prompt_embeds = torch.randn(batch, seq, cross_dim, generator=g)
    negative_prompt_embeds = torch.randn(batch, seq, cross_dim, generator=g)
    image = torch.rand(batch, 3, height, width, generator=g)
    mask = torch.zeros(batch, 3, height, width)
    mask[:, :, height // 4: height // 4 + height // 2, width // 4: width // 4 + width // 2] = 1.0   # 25 % hole
    image = image * (1.0 - mask)                                  # hole zeroed
    depth = torch.rand(batch, 1, height, width, generator=g) * 2.0 - 1.0
    latents = torch.randn(batch, latent_channels, hl, wl, generator=g)
    vae_noise = torch.randn(2 * batch, latent_channels, hl, wl, generator=g)   # uncond half, then cond half
    return dict(prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, image=image, mask=mask,
                depth=depth, latents=latents, vae_noise=vae_noise)
