"""Dataset-side transforms of the reference on the device: examples/brushnet/dataset/dataset.py.

`apply_transforms_depth` (:98-166), normalisation method "max_scene_depth": the scene depth is the maximum depth under the
mirror mask plus `delta` (or the given `max_scene_depth` without a mask), depth is clipped to [0, scene] and mapped to
[-1, 1] (or [0, 1]) — one masked max reduction and one streaming pass on the GPU (csrc/frontend.hip), no host round trip.
The "percentile" method (a sort) and the bicubic Resize / CenterCrop of torchvision (only active when the depth map is
not already `resolution` x `resolution`; SynMirror renders are 512 x 512) are not built and raise."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import hip


def apply_transforms_depth(depth_map, mask=None, normalization_method: str = "max_scene_depth", max_scene_depth: float = 5.0,
                           norm_range: Sequence[float] = (-1, 1), delta: float = 0.5, resolution: int = 512, device="cuda",
                           **kwargs) -> torch.Tensor:
    """Returns the [1, H, W] fp32 device tensor the reference's dataset hands to the collate function."""
    if normalization_method == "percentile":
        raise NotImplementedError("apply_transforms_depth: the 'percentile' normalisation (np.percentile) is not built")
    if normalization_method != "max_scene_depth":
        raise ValueError("Unsupported normalization method. Use 'percentile' or 'max_scene_depth'.")
    rng = [float(v) for v in norm_range]
    if rng not in ([0.0, 1.0], [-1.0, 1.0]):
        raise ValueError("Unsupported normalization range. Use [0, 1] or [-1, 1].")
    d = torch.as_tensor(np.ascontiguousarray(depth_map) if isinstance(depth_map, np.ndarray) else depth_map).to(device, torch.float32)
    if d.dim() != 2:
        raise ValueError("apply_transforms_depth takes an [H, W] depth map")
    if tuple(d.shape) != (resolution, resolution):
        raise NotImplementedError("apply_transforms_depth: torchvision's bicubic Resize + CenterCrop is not built; pass depth maps "
                                  f"of {resolution} x {resolution}")
    m = None
    if mask is not None:
        m = torch.as_tensor(np.ascontiguousarray(mask) if isinstance(mask, np.ndarray) else mask)
        if m.dim() == 3:
            m = m[:, :, 0]                                                    # dataset.py:111-112
        m = m.to(device, torch.float32).contiguous()
    out = hip.depth_normalize(d.contiguous(), m, max_scene_depth=max_scene_depth, delta=delta, signed_range=rng == [-1.0, 1.0])
    return out.unsqueeze(0)
