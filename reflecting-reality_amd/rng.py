"""Noise draws with the reference's generator semantics (utils/torch_utils.py randn_tensor): the draw happens on the
GENERATOR's device — a `torch.Generator('cuda')` (what examples/brushnet/test_brushnet.py:166 passes) draws on the GPU, a
CPU generator (or none) draws on the host and the result is uploaded; a list gives one generator per batch element."""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch


def randn_tensor(shape: Sequence[int], generator: Union[None, torch.Generator, List[torch.Generator]] = None,
                 device: Union[None, str, torch.device] = None) -> torch.Tensor:
    shape = tuple(shape)
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, (list, tuple)):
        if len(generator) == 1:
            generator = generator[0]
        else:
            if len(generator) != shape[0]:
                raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                                 f"effective batch size of {shape[0]}. Make sure the batch size matches the length of the "
                                 "generators.")
            return torch.cat([randn_tensor((1,) + shape[1:], g, device) for g in generator], dim=0)
    gdev = generator.device if generator is not None else torch.device("cpu")
    if gdev.type == "cuda" and device.type != "cuda":
        raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gdev.type}.")
    draw_on = gdev if generator is not None else torch.device("cpu")
    return torch.randn(shape, generator=generator, device=draw_on, dtype=torch.float32).to(device)
