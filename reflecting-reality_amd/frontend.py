"""Dataset-side transforms of the reference on the device: examples/brushnet/dataset/dataset.py.

`apply_transforms_depth` (:98-166): normalisation "max_scene_depth" (scene depth = maximum depth under the mirror mask +
`delta`, or the given `max_scene_depth`; one masked max reduction + one streaming pass) or "percentile" (clip to the
[2 %, 98 %] percentiles: a two-level radix select finds the four order statistics np.percentile interpolates between — no
sort), then torchvision's Resize(resolution, BICUBIC) + CenterCrop as ONE bicubic kernel evaluated inside the crop window.
`apply_transforms_normals` (:168-192, the map-valued modes): HWC -> CHW, the same resize / crop, Normalize([0.5], [0.5]).
All on the GPU (csrc/frontend.hip), no host round trip.

Bicubic: the reference pins torchvision 0.18 (MirrorFusion/README.md:34), where transforms.Resize defaults to
antialias=True and, for a tensor, hands it to torch.nn.functional.interpolate at EVERY scale (torchvision/transforms/
_functional_tensor.py `resize`): ATen's antialiased bicubic (Keys kernel a = -0.5 stretched by max(scale, 1), normalised
weights) — `antialias=None` / `True` here, mf_bicubic_aa_resize_crop.  `antialias=False` is the plain kernel (A = -0.75:
torchvision < 0.17's default for tensors), mf_bicubic_resize_crop.  SynMirror renders are 512 x 512 = `resolution`, where
Resize + CenterCrop are the identity in both.  The resize is checked against torch's own CPU interpolate (the arithmetic
torchvision calls; torchvision itself is absent from this image).  The reference MODULE imports h5py / torchvision / cv2 and
cannot be imported here, so the numpy statements around the resize (percentile / max-scene-depth normalisation) are checked
against the oracle's restatement of dataset.py:98-192 (oracle/mirrorfusion_ref.py, PARITY UNPINNED for those lines) — see
tests/test_frontend_gpu.py."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import hip


def _resize_geometry(h: int, w: int, resolution: int):
    """torchvision Resize(int): the smaller edge becomes `resolution`, the other keeps the aspect ratio (truncated);
    CenterCrop((resolution, resolution)): offsets int(round((size - resolution) / 2))."""
    if h <= w:
        nh, nw = resolution, int(resolution * w / h)
    else:
        nh, nw = int(resolution * h / w), resolution
    return (nh, nw), (int(round((nh - resolution) / 2.0)), int(round((nw - resolution) / 2.0)))


def _resize_crop(planes: torch.Tensor, resolution: int, antialias: Optional[bool], a: float = 1.0, b: float = 0.0) -> torch.Tensor:
    _, h, w = planes.shape
    (nh, nw), (top, left) = _resize_geometry(h, w, resolution)
    if (nh, nw) == (h, w) == (resolution, resolution):
        return planes if (a, b) == (1.0, 0.0) else hip.axpby_affine(planes, a, b)
    # antialias None = the reference's pinned torchvision (0.18): True, at every scale
    return hip.bicubic_resize_crop(planes, (nh, nw), (top, left), (resolution, resolution), a, b, antialias=antialias is not False)


def apply_transforms_depth(depth_map, mask=None, normalization_method: str = "max_scene_depth", max_scene_depth: float = 5.0,
                           norm_range: Sequence[float] = (-1, 1), delta: float = 0.5, resolution: int = 512, device="cuda",
                           antialias: Optional[bool] = None, **kwargs) -> torch.Tensor:
    """Returns the [1, resolution, resolution] fp32 device tensor the reference's dataset hands to the collate function."""
    if normalization_method not in ("percentile", "max_scene_depth"):
        raise ValueError("Unsupported normalization method. Use 'percentile' or 'max_scene_depth'.")
    rng = [float(v) for v in norm_range]
    if rng not in ([0.0, 1.0], [-1.0, 1.0]):
        raise ValueError("Unsupported normalization range. Use [0, 1] or [-1, 1].")
    d = torch.as_tensor(np.ascontiguousarray(depth_map) if isinstance(depth_map, np.ndarray) else depth_map).to(device, torch.float32)
    if d.dim() != 2:
        raise ValueError("apply_transforms_depth takes an [H, W] depth map")
    signed = rng == [-1.0, 1.0]
    if normalization_method == "percentile":
        out = hip.depth_percentile_normalize(d.contiguous(), signed_range=signed)                  # :115-127
    else:
        m = None
        if mask is not None:
            m = torch.as_tensor(np.ascontiguousarray(mask) if isinstance(mask, np.ndarray) else mask)
            if m.dim() == 3:
                m = m[:, :, 0]                                                    # dataset.py:111-112
            m = m.to(device, torch.float32).contiguous()
        out = hip.depth_normalize(d.contiguous(), m, max_scene_depth=max_scene_depth, delta=delta, signed_range=signed)
    return _resize_crop(out.unsqueeze(0), resolution, antialias)                                    # :150-164


def apply_transforms_normals(normals_map, resolution: int = 512, mask=None, normals_conditioning_mode: str = "concat", device="cuda",
                             antialias: Optional[bool] = None, **kwargs) -> torch.Tensor:
    """dataset.py:168-192 for the map-valued modes: [H, W, 3] -> [3, resolution, resolution], (x - 0.5) / 0.5.  The
    'ip_adapter' mode (one mean normal vector for the image encoder) belongs to the IP-Adapter path, which is out of scope."""
    if normals_conditioning_mode == "ip_adapter":
        raise NotImplementedError("normals_conditioning_mode='ip_adapter' (SURVEY.md §2 #14: IP-Adapter is outside the hot path)")
    x = torch.as_tensor(np.ascontiguousarray(normals_map) if isinstance(normals_map, np.ndarray) else normals_map).to(device, torch.float32)
    if x.dim() != 3 or x.shape[-1] != 3:
        raise ValueError("apply_transforms_normals takes an [H, W, 3] normals map")
    h, w, _ = x.shape
    if (h, w) == (resolution, resolution):
        return hip.hwc_to_chw_affine(x, 2.0, -1.0)                                                   # Normalize([0.5], [0.5])
    return _resize_crop(hip.hwc_to_chw_affine(x, 1.0, 0.0), resolution, antialias, 2.0, -1.0)
