"""The inference harness of examples/brushnet/test_brushnet.py on the HIP pipeline: the sample list is split statically
between the ranks (accelerate's `PartialState().split_between_processes`, :163-168), every rank runs its samples through
its own pipeline replica with ONE generator seeded once per rank (`torch.Generator("cuda").manual_seed(args.seed)`, :166)
and draws `num_images_per_validation` images per sample from it in sequence (:247-266); checkpoints are enumerated the
way `--all_ckpt` does (:271-285).  No data-path collective: each image is an independent unit."""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch

from . import distributed as D


def list_checkpoints(brushnet_path: str, ckpt_modulo: Optional[int] = None) -> List[str]:
    """`checkpoint-N` folders in ascending N, optionally every `ckpt_modulo` steps (test_brushnet.py:271-285)."""
    cps = [d for d in os.listdir(brushnet_path) if d.startswith("checkpoint")]
    cps = sorted(cps, key=lambda x: int(x.split("-")[1]))
    if ckpt_modulo is not None:
        cps = [c for c in cps if int(c.split("-")[1]) % ckpt_modulo == 0]
    return [os.path.join(brushnet_path, c) for c in cps]


def run_sharded(pipe, samples: Sequence[Dict], *, seed: int = 0, num_images_per_validation: int = 4,
                num_inference_steps: int = 50, guidance_scale: float = 7.5, brushnet_conditioning_scale: float = 1.0,
                output_type: str = "pt", rank: Optional[int] = None, world: Optional[int] = None,
                on_result: Optional[Callable[[int, List], None]] = None, generator_device: Optional[str] = None) -> Dict[int, List]:
    """Runs this rank's share of `samples` (dicts of pipeline kwargs: image, mask, depth / normals, prompt_embeds, ...).
    Returns {sample index: [images]} for the samples this rank owns; `on_result(index, images)` is called as each sample
    finishes (the script saves its image grid there)."""
    if rank is None or world is None:
        rank, world, _ = D.env_rank_world()
    lo, hi = D.shard_range(len(samples), rank, world)
    gdev = generator_device or (str(pipe.device) if torch.device(pipe.device).type == "cuda" else "cpu")
    generator = torch.Generator(gdev).manual_seed(seed)                      # one generator per rank, drawn from in sequence
    out: Dict[int, List] = {}
    for i in range(lo, hi):
        kw = dict(samples[i])
        images = []
        for _ in range(num_images_per_validation):
            res = pipe(num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, generator=generator,
                       brushnet_conditioning_scale=float(brushnet_conditioning_scale), output_type=output_type, **kw)
            images.append(res.images[0])
        out[i] = images
        if on_result is not None:
            on_result(i, images)
    return out
